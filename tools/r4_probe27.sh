#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r4p27; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "fan_in or sums_fused or upsample" > $O/t1.txt 2>&1; tail -6 $O/t1.txt | cut -c1-300
python -m pytest tests/test_bf16_gpu.py -m gpu -q -x -k "sums_fused" > $O/t1b.txt 2>&1; tail -2 $O/t1b.txt | cut -c1-300
python -m pytest tests/test_step_gpu.py tests/test_schedule_gpu.py tests/test_nets_gpu.py -m gpu -q -x > $O/t2.txt 2>&1; tail -5 $O/t2.txt | cut -c1-300
B="python bench.py --no-cpu-baseline --no-elided --no-bf16 --no-roofline --no-small --steps 8 --warmup 3"
for v in 0 1 0 1; do for c in 2; do echo -n "c$c FUSE_JOIN=$v: "; SSCG_FUSE_JOIN=$v $B --config $c 2>/dev/null | python -c "
import sys, json
t = sys.stdin.read(); i = t.index('{\"metric\"'); d = json.JSONDecoder().raw_decode(t[i:])[0]; print(d['ms_per_step'], d['host_issue_ms_per_step'], 'finite' if d['config']['losses_finite'] else 'NON-FINITE')"; done; done 2>&1 | tee $O/ab.txt
echo -n "c3: "; $B --config 3 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1
