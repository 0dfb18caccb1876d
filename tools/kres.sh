#!/bin/bash
# register / LDS / spill figures of every kernel in one object file (usage: tools/kres.sh <file.o> [name filter])
O=$(realpath $1); T=$(mktemp -d); cd $T
/opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.hip_fatbin $O fat.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=fat.bin --output=k.co --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes k.co | grep -E "^ +\.name:|\.vgpr_count|\.agpr_count|\.sgpr_count|group_segment_fixed|private_segment_fixed|spill_count" | sed 's/ \+/ /g' | paste - - - - - - - - | sed 's/\.group_segment_fixed_size/lds/; s/\.private_segment_fixed_size/scratch/' | c++filt | grep "${2:-.}" | cut -c1-300
rm -rf $T
