#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r4p19; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -m gpu -q -x > $O/ktests.txt 2>&1; tail -3 $O/ktests.txt
B="python bench.py --no-cpu-baseline --no-elided --no-bf16 --no-roofline --no-small --steps 8 --warmup 3"
for v in 0 1 0 1; do echo -n "c2 FUSE_BSUMS=$v: "; SSCG_FUSE_BSUMS=$v $B 2>/dev/null | python -c "
import sys, json
t = sys.stdin.read(); i = t.index('{\"metric\"'); d = json.JSONDecoder().raw_decode(t[i:])[0]; print(d['ms_per_step'], d['host_issue_ms_per_step'], 'finite' if d['config']['losses_finite'] else 'NON-FINITE')"; done 2>&1 | tee $O/bsums_c2.txt
