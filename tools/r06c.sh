export TMPDIR=/tmp
python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "norm or pixel or wgrad or head" > gpurun_out/r06c_tests.txt 2>&1; tail -6 gpurun_out/r06c_tests.txt
python -m pytest tests/test_schedule_gpu.py -m gpu -q -x -k "ordering" > gpurun_out/r06c_racecheck.txt 2>&1; tail -3 gpurun_out/r06c_racecheck.txt
SSCG_NORM_SLAB=0 python tools/norm_bench.py gpurun_out/r06c_norm_f32_flat.txt f32
python tools/norm_bench.py gpurun_out/r06c_norm_f32_slab.txt f32
SSCG_NORM_SLAB=0 python tools/norm_bench.py gpurun_out/r06c_norm_bf16_flat.txt bf16
python tools/norm_bench.py gpurun_out/r06c_norm_bf16_slab.txt bf16
OFF="SSCG_NORM_SLAB=0 SSCG_WGRAD_REDUCE_W4=0 SSCG_FUSE_FRONT=0"
tools/ab.sh "$OFF" "SSCG_WGRAD_REDUCE_W4=0 SSCG_FUSE_FRONT=0" "SSCG_NORM_SLAB=0 SSCG_FUSE_FRONT=0" "SSCG_NORM_SLAB=0 SSCG_WGRAD_REDUCE_W4=0" "X=1" > gpurun_out/r06c_ab.txt 2>&1
cat gpurun_out/r06c_ab.txt
