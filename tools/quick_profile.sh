#!/bin/bash
# bench line + per-shape table + kernel statistics of one configuration, no PMC passes (usage: tools/quick_profile.sh <tag> <config>)
set -u
TAG=$1; CFG=${2:-2}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
SSCG_BENCH_SHAPES=$OUT/${TAG}_conv_shapes_c$CFG.txt timeout 600 python $ROOT/bench.py --config $CFG > $OUT/${TAG}_bench_line_c$CFG.json 2> $OUT/${TAG}_bench_c$CFG.err
rm -rf /tmp/prof_$TAG; mkdir -p /tmp/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG/kt -o kt -- python $ROOT/bench.py --config $CFG --no-cpu-baseline --no-elided --no-bf16 --no-small --no-unblocked --steps 4 --warmup 2 > $OUT/${TAG}_kt_c$CFG.log 2>&1
DB=$(find /tmp/prof_$TAG/kt -name "*.db" | head -1)
python $ROOT/tools/kstats.py $DB $OUT/${TAG}_bench_kernel_stats_c$CFG.csv > $OUT/${TAG}_kstats_c$CFG.txt 2>&1
tail -2 $OUT/${TAG}_kstats_c$CFG.txt
cut -c1-700 $OUT/${TAG}_bench_line_c$CFG.json
