#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r4p8; mkdir -p $O
python -m pytest tests -m gpu -q -x --durations=25 > $O/suite.txt 2>&1; tail -32 $O/suite.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; tail -c 1500 $O/bench_c2.json
