#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
O=$ROOT/gpurun_out/r4p23; mkdir -p $O
cd $ROOT; python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "one_launch" 2>&1 | tail -2
cd /tmp; rm -rf /tmp/p23; mkdir -p /tmp/p23
timeout 600 rocprofv3 --kernel-trace -d /tmp/p23/kt -o kt -- python $ROOT/bench.py --config 2 --no-cpu-baseline --no-elided --no-bf16 --no-small --no-roofline --steps 4 --warmup 2 > $O/kt.log 2>&1
DB=$(find /tmp/p23/kt -name "*.db" | head -1)
for pat in finalize_conv_stats col_reduce_kernel norm_apply_kernel norm_bwd_apply wgrad_reduce addv_kernel finalize_bwd split_reduce_stats ks_reduce split3_kernel copyBuffer; do python $ROOT/tools/kslow.py $DB $pat 8; done > $O/kslow.txt 2>&1
python $ROOT/tools/gpu_idle.py $DB > $O/idle.txt 2>&1
head -60 $O/kslow.txt | cut -c1-200
