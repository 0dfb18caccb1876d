#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r4p31; mkdir -p $O
python -m pytest tests/test_bf16_gpu.py -m gpu -q -x -k "fan_in or conv_bf16 or sums_fused" 2>&1 | tail -4 | cut -c1-250
python -m pytest tests/test_bf16_gpu.py -m gpu -q -x -k "full_size or step" 2>&1 | tail -2 | cut -c1-250
B="python bench.py --no-cpu-baseline --no-elided --no-bf16 --no-roofline --no-small --steps 8 --warmup 3"
for v in 0 1 0 1; do echo -n "c3 FUSE_JOIN=$v: "; SSCG_FUSE_JOIN=$v $B --config 3 2>/dev/null | python -c "
import sys, json
t = sys.stdin.read(); i = t.index('{\"metric\"'); d = json.JSONDecoder().raw_decode(t[i:])[0]; print(d['ms_per_step'], d['host_issue_ms_per_step'], 'finite' if d['config']['losses_finite'] else 'NON-FINITE')"; done 2>&1 | tee $O/ab.txt
