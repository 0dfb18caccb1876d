#!/bin/bash
# step-level A/B of stream priorities (same box): side lanes low / fork lane high against the default
cd "$(dirname "$0")/.."
B="python bench.py --no-cpu-baseline --no-elided --no-bf16 --no-small --no-roofline --steps 12 --warmup 4"
run() { echo "== $1"; env $1 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"; }
run "SSCG_SIDE_PRIORITY=0"
run "SSCG_SIDE_PRIORITY=1"
run "SSCG_FORK_PRIORITY=-1"
run "SSCG_SIDE_PRIORITY=1 SSCG_FORK_PRIORITY=-1"
run "SSCG_SIDE_PRIORITY=0"
run "SSCG_SIDE_PRIORITY=1 SSCG_SIDE_LANES=3"
