#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r4p11; mkdir -p $O
WGF_SPLITS=8,12,16,24,32,48 timeout 600 python tools/wgrad1x1_bench.py > $O/wgf_splits.txt 2>&1; grep -v amdgpu $O/wgf_splits.txt | cut -c1-330
python -m pytest tests/test_dp_step_gpu.py tests/test_kernels_gpu.py tests/test_nets_gpu.py tests/test_parity_gpu.py tests/test_rccl_gpu.py tests/test_schedule_gpu.py tests/test_step_gpu.py -m gpu -q -x --durations=12 > $O/suite_rest.txt 2>&1; tail -18 $O/suite_rest.txt
