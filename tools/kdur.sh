#!/bin/bash
# true kernel durations (rocprofv3 --kernel-trace) of tools/convs_bench.py on chosen shapes: tools/kdur.sh <tag> "<SHAPE_IDX>" ["ENV=..."]
TAG=$1; IDX=$2; ENVS=${3:-}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/kd_$TAG; mkdir -p /tmp/kd_$TAG
env $ENVS SHAPE_IDX=$IDX timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kd_$TAG -o kt -- python $ROOT/tools/convs_bench.py time 0 > /tmp/kd_$TAG/log.txt 2>&1
DB=$(find /tmp/kd_$TAG -name "*.db" | head -1)
python - "$DB" <<'PY'
import sqlite3, sys, collections, re
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, (end-start) from kernels").fetchall()
agg = collections.defaultdict(list)
for n, d in rows:
    n = n.replace("(anonymous namespace)::", "")
    if "convs_kernel" in n or "g1x1" in n or "conv_kc" in n:
        agg[re.sub(r"\(.*", "", n)].append(d)
for k, v in sorted(agg.items()):
    v.sort()
    print("%-60s x%-4d median %7.1f us  min %7.1f" % (k, len(v), v[len(v) // 2] / 1e3, v[0] / 1e3))
PY
