#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r4p17; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py tests/test_bf16_gpu.py tests/test_step_gpu.py::test_variant_step_vs_oracle_golden -m gpu -q -x > $O/tests.txt 2>&1; tail -3 $O/tests.txt
B="python bench.py --no-cpu-baseline --no-elided --no-bf16 --no-roofline --no-small --steps 8 --warmup 3"
for c in 3 3 2; do echo -n "config $c: "; $B --config $c 2>/dev/null | python -c "
import sys, json
t = sys.stdin.read(); i = t.index('{\"metric\"'); d = json.JSONDecoder().raw_decode(t[i:])[0]; print(d['ms_per_step'], d['host_issue_ms_per_step'], 'finite' if d['config']['losses_finite'] else 'NON-FINITE')"; done 2>&1 | tee $O/bench.txt
