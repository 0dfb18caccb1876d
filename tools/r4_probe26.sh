#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
O=$ROOT/gpurun_out/r4p26; mkdir -p $O
cd $ROOT; python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "upsample or softmax or cross_entropy" 2>&1 | tail -2
python -m pytest tests/test_step_gpu.py -m gpu -q -x 2>&1 | tail -2
cd /tmp; rm -rf /tmp/p26; mkdir -p /tmp/p26
timeout 600 rocprofv3 --kernel-trace -d /tmp/p26/kt -o kt -- python $ROOT/bench.py --config 2 --no-cpu-baseline --no-elided --no-bf16 --no-small --no-roofline --steps 4 --warmup 2 > $O/kt.log 2>&1
DB=$(find /tmp/p26/kt -name "*.db" | head -1)
python - $DB $O/kernels.csv <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("PRAGMA table_info(kernels)").fetchall()]
g = [c for c in ("grid_x", "grid_size_x") if c in cols][0]
rows = db.execute("select start, end, stream_id, %s, name from kernels order by start" % g).fetchall()
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[4]]
lo = rows[adam[-5]][1]          # the last two steps
with open(sys.argv[2], "w") as f:
    for s, e, sid, gx, name in rows:
        if s >= lo:
            f.write("%d,%d,%s,%s,%s\n" % (s - lo, e - lo, sid, gx, name.replace(",", ";")[:70]))
PY
wc -l $O/kernels.csv; ls -la $O
cd $ROOT
B="python bench.py --no-cpu-baseline --no-elided --no-bf16 --no-roofline --no-small --steps 8 --warmup 3"
for c in 2 3; do echo -n "c$c: "; $B --config $c 2>/dev/null | python -c "
import sys, json
t = sys.stdin.read(); i = t.index('{\"metric\"'); d = json.JSONDecoder().raw_decode(t[i:])[0]; print(d['ms_per_step'], d['host_issue_ms_per_step'], 'finite' if d['config']['losses_finite'] else 'NON-FINITE')"; done
