#!/bin/bash
cd "$(dirname "$0")/.."
B="python bench.py --no-cpu-baseline --no-elided --no-bf16 --no-roofline --steps 6 --warmup 2"
run() { echo -n "$1 | $2: "; env $1 $B $2 2>/dev/null | python -c "
import sys, json
t = sys.stdin.read(); i = t.index('{\"metric\"'); d = json.JSONDecoder().raw_decode(t[i:])[0]; print(d['ms_per_step'], d['config']['losses_finite'], d.get('host_bound_case', {}).get('ms_per_step'))"; }
run "SSCG_X=0" ""
run "SSCG_X=0" "--no-small"
run "SSCG_BENCH_NO_EMPTY=1" ""
run "SSCG_SIDE_PRIORITY=1" ""
run "SSCG_SIDE_PRIORITY=0" ""
