#!/bin/bash
# round-4 probe 1: ordering checker on the shipped / failing schedules + NaN bisection at config-2 size
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r4p1; mkdir -p $O
( time timeout 600 python tests/aids/racecheck_step.py 3 64 2 1 ) > $O/rc_lanes2.txt 2>&1
( SSCG_SIDE_LANES=3 timeout 600 python tests/aids/racecheck_step.py 3 64 2 1 ) > $O/rc_lanes3.txt 2>&1
( SSCG_FORCE_DP=1 timeout 600 python tests/aids/racecheck_step.py 3 64 2 1 ) > $O/rc_dp.txt 2>&1
tail -5 $O/rc_lanes2.txt
{
tools/nan_probe.sh "SSCG_SIDE_LANES=3 SSCG_SIDE_PRIORITY=1" \
  "SSCG_SIDE_LANES=3 SSCG_SIDE_PRIORITY=1 SSCG_DBG_TOPWAIT=main,fork,side0,side1" \
  "SSCG_SIDE_LANES=3 SSCG_SIDE_PRIORITY=1 SSCG_DBG_TOPWAIT=main,fork" \
  "SSCG_SIDE_LANES=3 SSCG_SIDE_PRIORITY=1 SSCG_DBG_TOPWAIT=side0,side1" \
  "SSCG_SIDE_LANES=3 SSCG_SIDE_PRIORITY=1 SSCG_DBG_TOPWAIT=side0" \
  "SSCG_SIDE_LANES=3 SSCG_SIDE_PRIORITY=1 SSCG_DBG_TOPWAIT=side1"
} > $O/nan_bisect.txt 2>&1
cat $O/nan_bisect.txt
