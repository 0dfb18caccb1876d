#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r4p12; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "conv_fwd_bwd or adjoint or split_contraction or conv_transpose or conv_all_tile" > $O/ktests.txt 2>&1; tail -3 $O/ktests.txt
NB=$PWD/semi-supervised-segmentation-cyclegan_amd/libsscg_nobuf.so
SHAPE_IDX=0,3,4,8,9,10,11,1 timeout 600 python tools/convs_bench.py time 0 > $O/convs_buf.txt 2>&1
SSCG_LIB=$NB SHAPE_IDX=0,3,4,8,9,10,11,1 timeout 600 python tools/convs_bench.py time 0 > $O/convs_nobuf.txt 2>&1
paste -d'\n' <(grep -v amdgpu $O/convs_buf.txt | sed 's/^/buf   /') <(grep -v amdgpu $O/convs_nobuf.txt | sed 's/^/nobuf /')
B="python bench.py --no-cpu-baseline --no-elided --no-bf16 --no-roofline --no-small --steps 8 --warmup 3"
for v in "SSCG_LIB=$NB" "" "SSCG_LIB=$NB" ""; do echo -n "[${v:0:12}]: "; env $v $B 2>/dev/null | python -c "
import sys, json
t = sys.stdin.read(); i = t.index('{\"metric\"'); d = json.JSONDecoder().raw_decode(t[i:])[0]; print(d['ms_per_step'], d['host_issue_ms_per_step'], 'finite' if d['config']['losses_finite'] else 'NON-FINITE')"; done 2>&1 | tee $O/bench_buf.txt
