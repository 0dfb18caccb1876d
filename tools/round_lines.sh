#!/bin/bash
# The round's secondary bench lines on one box: config 3, config 5 per rank, forced data-parallel (RCCL at world size 1) beside the
# plain step, and the D-step kernel time with / without the fused PixelDiscriminator tail.  usage: tools/round_lines.sh <tag>
TAG=$1
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
[ -n "$SKIP_C3" ] || python bench.py --config 3 > $OUT/${TAG}_bench_line_c3.json 2> $OUT/${TAG}_c3.err
python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_bench_line_c5_per_rank.json 2> $OUT/${TAG}_c5.err
Q="--no-cpu-baseline --no-elided --no-bf16 --no-small --no-roofline --no-unblocked --steps 12 --warmup 4"
python bench.py $Q > $OUT/${TAG}_bench_no_dp_same_box.json 2>/dev/null
SSCG_FORCE_DP=1 NCCL_DEBUG=INFO python bench.py $Q 2>&1 | tee $OUT/${TAG}_bench_force_dp_rccl.log | grep "^{\"metric" > $OUT/${TAG}_bench_force_dp.json
SSCG_FORCE_DP=1 SSCG_DP_BUCKETS=0 python bench.py $Q 2>/dev/null | grep "^{\"metric" > $OUT/${TAG}_bench_force_dp_one_piece.json
python bench.py $Q > $OUT/${TAG}_bench_no_dp_same_box_2.json 2>/dev/null
for f in c3 c5_per_rank; do python -c "import json,sys; d=json.loads(open('$OUT/${TAG}_bench_line_$f.json').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['value'], d.get('host_bound_case',{}).get('ms_per_step'), d.get('roofline',{}).get('achieved'))"; done
for f in no_dp_same_box force_dp force_dp_one_piece no_dp_same_box_2; do python -c "import json; d=json.loads(open('$OUT/${TAG}_bench_$f.json').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'])"; done
grep -c "NCCL INFO" $OUT/${TAG}_bench_force_dp_rccl.log
