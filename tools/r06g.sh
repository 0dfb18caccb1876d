export TMPDIR=/tmp
cd /root/repo; python -m pytest tests -m gpu -q -x --durations=15 > gpurun_out/r06g_gpu_suite.txt 2>&1; tail -22 gpurun_out/r06g_gpu_suite.txt
tools/profile_round.sh r06 2 split > gpurun_out/r06_profile.log 2>&1; tail -3 gpurun_out/r06_profile.log | cut -c1-1500
