export TMPDIR=/tmp
python -m pytest tests/test_kernels_gpu.py -m gpu -q -s -k "pixel_discriminator_front and 256x256" > gpurun_out/r06d_front.txt 2>&1; grep -n "fused / unfused\|passed\|failed\|Error" gpurun_out/r06d_front.txt | cut -c1-400
for cfg in "4 0" "4 1" "4 2" "2 2"; do set -- $cfg; SSCG_NORM_SLAB_U8A=$1 SSCG_NORM_SLAB_U8B=$2 python tools/norm_bench.py gpurun_out/r06d_norm_bf16_a$1_b$2.txt bf16; done
AB_ARGS="--config 3" tools/ab.sh "SSCG_NORM_SLAB=0" "SSCG_NORM_SLAB_U8A=4 SSCG_NORM_SLAB_U8B=0" "SSCG_NORM_SLAB_U8A=4 SSCG_NORM_SLAB_U8B=2" "SSCG_NORM_SLAB_U8A=2 SSCG_NORM_SLAB_U8B=1" > gpurun_out/r06d_ab_c3.txt 2>&1
cat gpurun_out/r06d_ab_c3.txt
