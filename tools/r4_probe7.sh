#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r4p7; mkdir -p $O
timeout 600 python tools/wgrad1x1_bench.py > $O/wgrad1x1.txt 2>&1; grep -v amdgpu.ids $O/wgrad1x1.txt
python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "wgrad or split_contraction or adjoint or conv_fwd_bwd" > $O/ktests.txt 2>&1; tail -4 $O/ktests.txt
SHAPE_IDX=0,3,4,1 timeout 600 python tools/convs_bench.py time 0 0x204 0x203 0x404 > $O/convs_splitk.txt 2>&1; grep -v amdgpu.ids $O/convs_splitk.txt
B="python bench.py --no-cpu-baseline --no-elided --no-bf16 --no-roofline --no-small --steps 8 --warmup 3"
for v in "SSCG_WGRAD_TUNING=5" "" "SSCG_WGRAD_TUNING=5" ""; do echo -n "[$v]: "; env $v $B 2>/dev/null | python -c "
import sys, json
t = sys.stdin.read(); i = t.index('{\"metric\"'); d = json.JSONDecoder().raw_decode(t[i:])[0]; print(d['ms_per_step'], d['host_issue_ms_per_step'], 'finite' if d['config']['losses_finite'] else 'NON-FINITE')"; done 2>&1 | tee $O/bench_wgf.txt
