#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r4p4; mkdir -p $O
f() { grep -v "^\[W\|Warning\|amdgpu.ids" $1 | tail -${2:-9}; }
( SSCG_DBG_NO_COPY_SYNC=1 SSCG_FUZZ_DELAY_FORK=20000000 timeout 900 python tests/aids/fuzz_step.py 2 3 64 2 ) > $O/fuzz_nosync_delayfork.txt 2>&1; f $O/fuzz_nosync_delayfork.txt
( SSCG_FUZZ_DELAY_FORK=20000000 timeout 900 python tests/aids/fuzz_step.py 2 3 64 2 ) > $O/fuzz_fixed_delayfork.txt 2>&1; f $O/fuzz_fixed_delayfork.txt
( SSCG_DBG_NO_COPY_SYNC=1 timeout 900 python tests/aids/fuzz_step.py 6 3 64 2 ) > $O/fuzz_nosync.txt 2>&1; f $O/fuzz_nosync.txt
( timeout 900 python tests/aids/fuzz_step.py 6 3 64 2 ) > $O/fuzz_fixed.txt 2>&1; f $O/fuzz_fixed.txt
