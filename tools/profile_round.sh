#!/bin/bash
# Round profile of one bench configuration on the GPU box: bench line + per-shape table, rocprofv3 kernel trace -> per-kernel
# stats, four separate --pmc passes -> per-kernel counter means.  usage: tools/profile_round.sh <tag> <config> <family>
# (family = f32|bf16, the suffix bench.py's pmc_traffic() looks for).  Output under gpurun_out/<tag>_*; copy into profiles/.
set -u
TAG=$1; CFG=$2; FAM=$3
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
i=0
# FETCH_SIZE and WRITE_SIZE do not fit the TCC counter slots together: one pass each
for PMC in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d /tmp/prof_$TAG/pmc$i -o pmc -- python $ROOT/bench.py --config $CFG --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-elided --no-bf16 --no-small --no-unblocked > $OUT/${TAG}_pmc${i}_c$CFG.log 2>&1
done
python $ROOT/tools/pmc_summary.py $OUT/${TAG}_pmc_per_kernel_$FAM.json /tmp/prof_$TAG/pmc1 /tmp/prof_$TAG/pmc2 /tmp/prof_$TAG/pmc3 /tmp/prof_$TAG/pmc4 > $OUT/${TAG}_pmc_digest_c$CFG.txt 2>&1
tail -2 $OUT/${TAG}_kstats_c$CFG.txt
cat $OUT/${TAG}_bench_line_c$CFG.json
