#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r4p14; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "conv_fwd_bwd or adjoint or split_contraction" > $O/ktests.txt 2>&1; tail -2 $O/ktests.txt
# class 128x64 (code 4), never split (0x100): two stages (bit 16) vs three stages (bit 17); and the default plan
SHAPE_IDX=0,4,3,1 timeout 600 python tools/convs_bench.py time 0 0x10104 0x20104 2>&1 | grep -v amdgpu | tee $O/stages.txt
