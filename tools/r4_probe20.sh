#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r4p20; mkdir -p $O
python -m pytest tests/test_bf16_gpu.py -m gpu -q -x -s -k "fused_into or conv_bf16 or bottleneck" > $O/tests.txt 2>&1; grep -v "^$" $O/tests.txt | tail -12 | cut -c1-160
B="python bench.py --no-cpu-baseline --no-elided --no-bf16 --no-roofline --no-small --steps 8 --warmup 3 --config 3"
for v in 0 1 0 1; do echo -n "c3 FUSE_BSUMS=$v: "; SSCG_FUSE_BSUMS=$v $B 2>/dev/null | python -c "
import sys, json
t = sys.stdin.read(); i = t.index('{\"metric\"'); d = json.JSONDecoder().raw_decode(t[i:])[0]; print(d['ms_per_step'], d['host_issue_ms_per_step'], 'finite' if d['config']['losses_finite'] else 'NON-FINITE')"; done 2>&1 | tee $O/bsums_c3.txt
