export TMPDIR=/tmp
AB_ARGS="--config 3" tools/ab.sh "X=0" "SSCG_FUSE_BSUMS=1" > gpurun_out/r06f_ab_c3.txt 2>&1
cat gpurun_out/r06f_ab_c3.txt
SKIP_C3=1 tools/round_lines.sh r06f > gpurun_out/r06f_lines.log 2>&1; tail -8 gpurun_out/r06f_lines.log
python -c "
import json; d=json.load(open('gpurun_out/r06f_bench_force_dp.json')); print(d['ms_per_step'], d['rccl'], d['timed_region_alloc'])"
