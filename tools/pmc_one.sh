#!/bin/bash
# SQ counter passes over one command (tuning aid).  usage: tools/pmc_one.sh <tag> <command...>; output gpurun_out/<tag>_pmc.json
set -u
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/pmc_$TAG; mkdir -p /tmp/pmc_$TAG
i=0
for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_WAVES SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d /tmp/pmc_$TAG/p$i -o pmc -- "$@" > $OUT/${TAG}_pmc$i.log 2>&1
done
python $ROOT/tools/pmc_summary.py $OUT/${TAG}_pmc.json /tmp/pmc_$TAG/p1 /tmp/pmc_$TAG/p2 /tmp/pmc_$TAG/p3 > $OUT/${TAG}_pmc_digest.txt 2>&1
cat $OUT/${TAG}_pmc_digest.txt | head -20
