export TMPDIR=/tmp
cd /root/repo
python -m pytest tests/test_bf16_gpu.py tests/test_kernels_gpu.py -m gpu -q -x -s -k "sums" > gpurun_out/r06q_tests.txt 2>&1; tail -3 gpurun_out/r06q_tests.txt; grep -c "fused vs reduction" gpurun_out/r06q_tests.txt
SSCG_FUSE_BSUMS=1 python -m pytest tests/test_bf16_gpu.py -m gpu -q -x > gpurun_out/r06q_tests_fused.txt 2>&1; tail -3 gpurun_out/r06q_tests_fused.txt
AB_ARGS="--config 3" tools/ab.sh "X=1" "SSCG_FUSE_BSUMS=1" > gpurun_out/r06q_ab_c3.txt 2>&1; cat gpurun_out/r06q_ab_c3.txt
