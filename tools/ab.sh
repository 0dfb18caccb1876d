#!/bin/bash
# step-level A/B on one box: tools/ab.sh "ENV=a" "ENV=b" ...  (each setting twice, interleaved)
cd "$(dirname "$0")/.."
B="python bench.py --no-cpu-baseline --no-elided --no-bf16 --no-small --no-roofline --no-unblocked --steps 12 --warmup 4 ${AB_ARGS:-}"
run() { echo -n "$1: "; env $1 $B 2>/dev/null | python -c "
import sys, json
t = sys.stdin.read(); i = t.index('{\"metric\"'); d = json.JSONDecoder().raw_decode(t[i:])[0]; print(d['ms_per_step'], d['value'])"; }
for rep in 1 2; do for s in "$@"; do run "$s"; done; done
