export TMPDIR=/tmp
cd /root/repo
AB_ARGS="--config 3" tools/ab.sh "SSCG_SIDE_LANES=2" "SSCG_SIDE_LANES=1" "SSCG_SIDE_LANES=3" > gpurun_out/r06k_lanes_c3.txt 2>&1; cat gpurun_out/r06k_lanes_c3.txt
tools/ab.sh "X=0" "SSCG_KS_K12864=512" "SSCG_KS_T128_SHORT=1024" "SSCG_KS_T128_SHORT=256" "SSCG_KS_K128=256" "SSCG_KS_T12864=512" > gpurun_out/r06k_policy_c2.txt 2>&1; cat gpurun_out/r06k_policy_c2.txt
