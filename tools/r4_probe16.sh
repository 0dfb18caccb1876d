#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r4p16; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -m gpu -q -x > $O/ktests.txt 2>&1; tail -3 $O/ktests.txt
NS=$PWD/semi-supervised-segmentation-cyclegan_amd/libsscg_nostage.so
SHAPE_IDX=0,3,4,8,9,10,11,18,19,20,21 timeout 600 python tools/convs_bench.py time 0 > $O/convs_stage.txt 2>&1
SSCG_LIB=$NS SHAPE_IDX=0,3,4,8,9,10,11,18,19,20,21 timeout 600 python tools/convs_bench.py time 0 > $O/convs_nostage.txt 2>&1
paste -d'\n' <(grep -v amdgpu $O/convs_stage.txt | sed 's/^/stage   /') <(grep -v amdgpu $O/convs_nostage.txt | sed 's/^/nostage /')
B="python bench.py --no-cpu-baseline --no-elided --no-bf16 --no-roofline --no-small --steps 8 --warmup 3"
for v in "SSCG_LIB=$NS" "" "SSCG_LIB=$NS" ""; do echo -n "[${v:0:12}]: "; env $v $B 2>/dev/null | python -c "
import sys, json
t = sys.stdin.read(); i = t.index('{\"metric\"'); d = json.JSONDecoder().raw_decode(t[i:])[0]; print(d['ms_per_step'], d['host_issue_ms_per_step'], 'finite' if d['config']['losses_finite'] else 'NON-FINITE')"; done 2>&1 | tee $O/bench.txt
