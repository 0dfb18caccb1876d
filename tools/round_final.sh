#!/bin/bash
# evidence of the final tree of a round (usage: tools/round_final.sh <tag>, e.g. r05): profile of config 2 (bench line, shapes, kernel stats, PMC), secondary lines, full -m gpu suite
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=${1:-r05}
tools/profile_round.sh $TAG 2 split > gpurun_out/${TAG}_profile.log 2>&1; tail -3 gpurun_out/${TAG}_profile.log | cut -c1-600
tools/profile_round.sh ${TAG} 3 bf16 > gpurun_out/${TAG}_profile_c3.log 2>&1; tail -2 gpurun_out/${TAG}_profile_c3.log | cut -c1-300
SKIP_C3=1 tools/round_lines.sh ${TAG} > gpurun_out/${TAG}_lines.log 2>&1; tail -6 gpurun_out/${TAG}_lines.log
python -c 'import __graft_entry__ as g; g.smoke()' > gpurun_out/${TAG}_smoke.txt 2>&1; tail -1 gpurun_out/${TAG}_smoke.txt | cut -c1-400
cd /root/repo; python -m pytest tests -m gpu -q -x --durations=15 > gpurun_out/${TAG}_gpu_suite.txt 2>&1; tail -20 gpurun_out/${TAG}_gpu_suite.txt
