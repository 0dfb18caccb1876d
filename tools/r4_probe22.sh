#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r4p22; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "one_launch" > $O/t1.txt 2>&1; tail -3 $O/t1.txt | cut -c1-200
python -m pytest tests/test_schedule_gpu.py tests/test_step_gpu.py -m gpu -q -x > $O/t2.txt 2>&1; tail -4 $O/t2.txt | cut -c1-200
B="python bench.py --no-cpu-baseline --no-elided --no-bf16 --no-roofline --no-small --steps 8 --warmup 3"
for v in 0 1 0 1; do for c in 2 3; do echo -n "c$c BATCH_TRANSPOSES=$v: "; SSCG_BATCH_TRANSPOSES=$v $B --config $c 2>/dev/null | python -c "
import sys, json
t = sys.stdin.read(); i = t.index('{\"metric\"'); d = json.JSONDecoder().raw_decode(t[i:])[0]; print(d['ms_per_step'], d['host_issue_ms_per_step'], 'finite' if d['config']['losses_finite'] else 'NON-FINITE')"; done; done 2>&1 | tee $O/ab.txt
for v in 0 1 0 1; do echo -n "host-bound BATCH_TRANSPOSES=$v: "; SSCG_BATCH_TRANSPOSES=$v python bench.py --host-bound-only 2>/dev/null | tail -1 | cut -c1-300; done 2>&1 | tee $O/hb.txt
