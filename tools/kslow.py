#!/usr/bin/env python
"""Longest dispatches of the kernels matching a pattern in a rocprofv3 --kernel-trace database, grouped by (grid, workgroup): where a
family of small kernels hides a few long launches.  usage: python tools/kslow.py <results.db> <pattern> [rows]"""
import sqlite3
import sys
from collections import defaultdict


def main():
    db = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in db.execute("PRAGMA table_info(kernels)").fetchall()]
    g = [c for c in ("grid_x", "grid_size_x") if c in cols][0]
    gy = [c for c in ("grid_y", "grid_size_y") if c in cols][0]
    wg = [c for c in ("workgroup_x", "workgroup_size_x") if c in cols][0]
    rows = db.execute("select name, %s, %s, %s, end - start from kernels where name like ?" % (g, gy, wg), ("%" + sys.argv[2] + "%",)).fetchall()
    by = defaultdict(list)
    for name, gx, gyv, w, d in rows:
        by[(name[:60], gx, gyv, w)].append(d)
    tot = sum(sum(v) for v in by.values())
    top = sorted(by.items(), key=lambda kv: -sum(kv[1]))[:int(sys.argv[3]) if len(sys.argv) > 3 else 12]
    print("%s: %d dispatches, %.2f ms total" % (sys.argv[2], len(rows), tot / 1e6))
    for (name, gx, gyv, w), v in top:
        print("  %6.2f ms %5.1f%%  x%-4d avg %8.1f us max %8.1f us  grid %s x %s wg %s  %s" % (sum(v) / 1e6, 100.0 * sum(v) / tot, len(v), sum(v) / len(v) / 1e3,
                                                                                       max(v) / 1e3, gx, gyv, w, name))


if __name__ == "__main__":
    main()
