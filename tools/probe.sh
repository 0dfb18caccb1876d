#!/bin/bash
# ONE parameterised probe for the GPU box (replaces the round-4 one-off scripts): optional pytest selection, then an interleaved
# step-level A/B of environment settings (each setting twice; boxes differ by ~4 %, so only same-box A/Bs are comparable).
#   tools/probe.sh <name> [-t "<pytest args>"] [-c "<configs>"] [-b "<extra bench.py args>"] [--] "ENV=a ENV2=b" "" "SSCG_LIB=..." ...
# An empty string is the baseline setting.  Output: gpurun_out/<name>/{tests.txt,ab.txt}.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
NAME=$1; shift
TESTS=""; CONFIGS="2"; BARGS=""
while [ $# -gt 0 ]; do
  case "$1" in
    -t) TESTS=$2; shift 2;;
    -c) CONFIGS=$2; shift 2;;
    -b) BARGS=$2; shift 2;;
    --) shift; break;;
    *) break;;
  esac
done
O=gpurun_out/$NAME; mkdir -p $O
if [ -n "$TESTS" ]; then
  timeout ${PROBE_TEST_TIMEOUT:-1500} python -m pytest $TESTS -m gpu -q -x > $O/tests.txt 2>&1; tail -3 $O/tests.txt
fi
B="python bench.py --no-cpu-baseline --no-elided --no-bf16 --no-roofline --no-small --no-unblocked --steps ${PROBE_STEPS:-8} --warmup 3 $BARGS"
[ $# -eq 0 ] && set -- ""
for c in $CONFIGS; do
  for rep in 1 2; do
    for s in "$@"; do
      echo -n "c$c [${s:-baseline}]: "
      env $s timeout 600 $B --config $c 2>/dev/null | python -c "
import sys, json
t = sys.stdin.read()
try:
    i = t.index('{\"metric\"'); d = json.JSONDecoder().raw_decode(t[i:])[0]
    print(d['ms_per_step'], d['host_issue_ms_per_step'], 'finite' if d['config']['losses_finite'] else 'NON-FINITE')
except Exception as e:
    print('FAILED', type(e).__name__)"
    done
  done
done 2>&1 | tee $O/ab.txt
