#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r4p30; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "fan_in or sums_fused" 2>&1 | tail -2
python -m pytest tests/test_accuracy_gpu.py -m gpu -q -x -s > $O/accuracy.txt 2>&1; grep -E "pooled|passed|failed|rms|DeepLab|forward" $O/accuracy.txt | cut -c1-300 | head -12
B="python bench.py --no-cpu-baseline --no-elided --no-bf16 --no-roofline --no-small --steps 8 --warmup 3"
L=$PWD/semi-supervised-segmentation-cyclegan_amd/libsscg_rowwise.so
for v in "SSCG_LIB=$L" "" "SSCG_LIB=$L" ""; do echo -n "c2 [${v:0:8}]: "; env $v $B --config 2 2>/dev/null | python -c "
import sys, json
t = sys.stdin.read(); i = t.index('{\"metric\"'); d = json.JSONDecoder().raw_decode(t[i:])[0]; print(d['ms_per_step'], d['host_issue_ms_per_step'], 'finite' if d['config']['losses_finite'] else 'NON-FINITE')"; done 2>&1 | tee $O/ab.txt
