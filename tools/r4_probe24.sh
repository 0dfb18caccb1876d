#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r4p24; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "one_launch or norm_act or conv_fwd_bwd" > $O/t1.txt 2>&1; tail -3 $O/t1.txt | cut -c1-200
python -m pytest tests/test_step_gpu.py tests/test_nets_gpu.py -m gpu -q -x > $O/t2.txt 2>&1; tail -4 $O/t2.txt | cut -c1-200
B="python bench.py --no-cpu-baseline --no-elided --no-bf16 --no-roofline --no-small --steps 8 --warmup 3"
for lib in base new base new; do for c in 2 3; do echo -n "c$c lib=$lib: "; SSCG_LIB=$( [ $lib = base ] && echo $PWD/semi-supervised-segmentation-cyclegan_amd/libsscg_fcs16.so ) $B --config $c 2>/dev/null | python -c "
import sys, json
t = sys.stdin.read(); i = t.index('{\"metric\"'); d = json.JSONDecoder().raw_decode(t[i:])[0]; print(d['ms_per_step'], d['host_issue_ms_per_step'], 'finite' if d['config']['losses_finite'] else 'NON-FINITE')"; done; done 2>&1 | tee $O/ab.txt
