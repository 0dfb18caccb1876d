#!/bin/bash
# config 3 (bf16) kernel statistics of the final tree: rocprofv3 --kernel-trace --stats of the bench command, per-kernel table
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/prof_c3; mkdir -p /tmp/prof_c3
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_c3/kt -o kt -- python $ROOT/bench.py --config 3 --no-cpu-baseline --no-elided --no-bf16 --no-small --steps 4 --warmup 2 > $OUT/r04_kt_c3.log 2>&1
DB=$(find /tmp/prof_c3/kt -name "*.db" | head -1)
python $ROOT/tools/kstats.py $DB $OUT/r04_bench_kernel_stats_c3.csv > $OUT/r04_kstats_c3.txt 2>&1
head -25 $OUT/r04_kstats_c3.txt | cut -c1-150; tail -1 $OUT/r04_kstats_c3.txt
grep -o '"ms_per_step": [0-9.]*' $OUT/r04_kt_c3.log | head -1
