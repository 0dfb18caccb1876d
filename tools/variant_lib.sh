#!/bin/bash
# Build an ablation variant of the library next to the product one: tools/variant_lib.sh <suffix> <file.hip> "-DKNOB=0 ..."  ->
# semi-supervised-segmentation-cyclegan_amd/libsscg_<suffix>.so (run with SSCG_LIB=<path>; git-ignored, travels to the GPU box).
set -e
SUF=$1; SRC=$2; DEFS=$3
CS=$(cd "$(dirname "$0")/../semi-supervised-segmentation-cyclegan_amd/csrc" && pwd)
cd $CS
EXTRA=""; [ "$SRC" = "conv_split.hip" ] && EXTRA="-fno-slp-vectorize"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}; ARCH=${ARCH:-gfx950}
OBJ=$(mktemp -d)/variant_$SUF.o          # (private: concurrent variant builds do not collide)
$HIPCC --offload-arch=$ARCH -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $EXTRA $DEFS -c $SRC -o $OBJ
OBJS=""
for f in conv_igemm conv_bf16 conv_split conv_thin conv_wgrad norm pointwise loss_optim; do
  if [ "$f.hip" = "$SRC" ]; then OBJS="$OBJS $OBJ"; else OBJS="$OBJS $f.o"; fi
done
$HIPCC --offload-arch=$ARCH -shared -fPIC -o ../libsscg_$SUF.so $OBJS
ls -la ../libsscg_$SUF.so
