export TMPDIR=/tmp
python -m pytest tests/test_kernels_gpu.py -m gpu -q -s -k "pixel_discriminator_front and 256x256" > gpurun_out/r06e_front.txt 2>&1; grep -n "fused / unfused\|rel-L2\|passed\|failed\|Error" gpurun_out/r06e_front.txt | cut -c1-400
AB_ARGS="--config 3" tools/ab.sh "X=0" "SSCG_RED_U8=2" "SSCG_RED_U8=4" "SSCG_NORM_SLAB_U8B=2" > gpurun_out/r06e_ab_c3.txt 2>&1
cat gpurun_out/r06e_ab_c3.txt
tools/ab.sh "X=0" "SSCG_NORM_SLAB_U4A=2" "SSCG_NORM_SLAB_U4B=2" "SSCG_NORM_SLAB_U4A=2 SSCG_NORM_SLAB_U4B=2" > gpurun_out/r06e_ab_c2.txt 2>&1
cat gpurun_out/r06e_ab_c2.txt
