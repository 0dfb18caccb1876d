#!/bin/bash
# Does a configuration of the side lanes keep the losses finite?  (profiles/r03_experiments.txt item 12)  usage: tools/nan_probe.sh "ENV=..." ...
cd "$(dirname "$0")/.."
B="python bench.py --no-cpu-baseline --no-elided --no-bf16 --no-roofline --no-small --steps 6 --warmup 2"
for s in "$@"; do echo -n "$s: "; env $s $B 2>/dev/null | python -c "
import sys, json
t = sys.stdin.read(); i = t.index('{\"metric\"'); d = json.JSONDecoder().raw_decode(t[i:])[0]; print(d['ms_per_step'], 'finite' if d['config']['losses_finite'] else 'NON-FINITE')"; done
