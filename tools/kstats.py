#!/usr/bin/env python
"""Per-kernel summary (calls, total, average, share) of a rocprofv3 --kernel-trace results database.
usage: python tools/kstats.py <results.db> [out.csv]   - per-step figures assume 2 adam_kernel launches per step."""
import collections
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, (end-start) from kernels").fetchall()
    agg = collections.defaultdict(lambda: [0, 0, 10 ** 12, 0])
    for n, d in rows:
        n = n.replace("(anonymous namespace)::", "")
        n = re.sub(r"^void ", "", n)
        n = re.sub(r"\(.*", "", n) if n.startswith("at::") else re.sub(r"\((KcParams|WgParams|[A-Za-z_ ]*\*?,? ?)[^<>]*\)$", "", n)
        a = agg[n]
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    tot = sum(v[1] for v in agg.values())
    out = sorted(agg.items(), key=lambda kv: -kv[1][1])
    adam = [v[0] for k, v in agg.items() if "adam" in k]
    steps = adam[0] / 2 if adam else 1
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as f:
            f.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"\n')
            for n, (k, d, mn, mx) in out:
                f.write('"%s",%d,%d,%.1f,%.2f,%d,%d\n' % (n, k, d, d / k, 100 * d / tot, mn, mx))
    for n, (k, d, mn, mx) in out[:40]:
        print("%7.2f ms/step %5.1f%% x%-6.1f avg %7.1f us  %s" % (d / steps / 1e6, 100 * d / tot, k / steps, d / k / 1e3, n[:100]))
    print("total kernel time %.1f ms/step over %g steps" % (tot / steps / 1e6, steps))


if __name__ == "__main__":
    main()
