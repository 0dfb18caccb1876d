#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r4p5; mkdir -p $O
python -m pytest tests -m gpu -q -x --durations=120 > $O/suite.txt 2>&1; tail -5 $O/suite.txt
tools/nan_probe.sh "SSCG_SIDE_LANES=3 SSCG_SIDE_PRIORITY=1" "SSCG_FORCE_DP=1 SSCG_SIDE_LANES=3" "SSCG_SIDE_LANES=3 SSCG_SIDE_PRIORITY=1" "SSCG_FORCE_DP=1 SSCG_SIDE_LANES=3" "SSCG_FORCE_DP=1" "" > $O/nan_probe.txt 2>&1; cat $O/nan_probe.txt
