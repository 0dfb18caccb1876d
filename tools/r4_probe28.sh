#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r4p28; mkdir -p $O
L=$PWD/semi-supervised-segmentation-cyclegan_amd/libsscg_fcs16.so
{ python tools/smoke_report.py; SSCG_FUSE_HEAD=0 python tools/smoke_report.py; SSCG_LIB=$L python tools/smoke_report.py; SSCG_LIB=$L SSCG_FUSE_HEAD=0 python tools/smoke_report.py; } 2>/dev/null | grep "^\[" | tee $O/smoke_variants.txt
python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -1 | cut -c1-500 | tee $O/smoke.txt
bash tools/r4_c3_profile.sh
