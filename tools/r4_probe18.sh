#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r4p18; mkdir -p $O
NS=$PWD/semi-supervised-segmentation-cyclegan_amd/libsscg_b16ns.so
B="python bench.py --no-cpu-baseline --no-elided --no-bf16 --no-roofline --no-small --steps 8 --warmup 3 --config 3"
for v in "SSCG_LIB=$NS" "" "SSCG_LIB=$NS" ""; do echo -n "c3 [${v:0:12}]: "; env $v $B 2>/dev/null | python -c "
import sys, json
t = sys.stdin.read(); i = t.index('{\"metric\"'); d = json.JSONDecoder().raw_decode(t[i:])[0]; print(d['ms_per_step'], d['host_issue_ms_per_step'], 'finite' if d['config']['losses_finite'] else 'NON-FINITE')"; done 2>&1 | tee $O/bench.txt
