export TMPDIR=/tmp
cd /root/repo
python -m pytest tests/test_kernels_gpu.py tests/test_bf16_gpu.py tests/test_nets_gpu.py -m gpu -q -x > gpurun_out/r06j_tests.txt 2>&1; tail -4 gpurun_out/r06j_tests.txt
PREV=/root/repo/tools/experiments/libsscg_prev.so
tools/ab.sh "SSCG_LIB=$PREV" "X=1" > gpurun_out/r06j_ab_c2.txt 2>&1; cat gpurun_out/r06j_ab_c2.txt
AB_ARGS="--config 3" tools/ab.sh "SSCG_LIB=$PREV" "X=1" > gpurun_out/r06j_ab_c3.txt 2>&1; cat gpurun_out/r06j_ab_c3.txt
