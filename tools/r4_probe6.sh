#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r4p6; mkdir -p $O
python -m pytest tests/test_accuracy_gpu.py tests/test_step_gpu.py::test_variant_step_vs_oracle_golden tests/test_step_gpu.py::test_first_step_other_datasets_vs_oracle_golden tests/test_bf16_gpu.py::test_cityscapes_first_step_bf16_vs_fp64_oracle tests/test_schedule_gpu.py::test_pool_swap_runs_stay_finite -m gpu -q -s --durations=10 > $O/new_tests.txt 2>&1; tail -60 $O/new_tests.txt
B="python bench.py --no-cpu-baseline --no-elided --no-bf16 --no-roofline --no-small --steps 8 --warmup 3"
for v in 0 1 0 1; do echo -n "STACK_GIS=$v: "; SSCG_STACK_GIS=$v $B 2>/dev/null | python -c "
import sys, json
t = sys.stdin.read(); i = t.index('{\"metric\"'); d = json.JSONDecoder().raw_decode(t[i:])[0]; print(d['ms_per_step'], d['host_issue_ms_per_step'], 'finite' if d['config']['losses_finite'] else 'NON-FINITE')"; done 2>&1 | tee $O/stack_gis.txt
python tests/aids/flake_pool.py 20 1 > $O/flake20.txt 2>&1; tail -2 $O/flake20.txt
