#!/usr/bin/env python
"""Per-dispatch rows (kernel, grid, workgroup, duration) of a rocprofv3 --kernel-trace database, for kernels whose name contains a
pattern: effective bandwidth of the memory-bound kernels launch by launch.  usage: python tools/kdispatch.py <results.db> <pattern> [out.csv]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in db.execute("PRAGMA table_info(kernels)").fetchall()]
    print("columns:", cols)
    want = [c for c in ("name", "start", "end", "grid_x", "grid_size_x", "workgroup_x", "workgroup_size_x", "grid_y", "grid_size_y", "grid_z", "grid_size_z") if c in cols]
    rows = db.execute("select %s from kernels where name like ? order by start" % ", ".join(want), ("%" + sys.argv[2] + "%",)).fetchall()
    out = open(sys.argv[3], "w") if len(sys.argv) > 3 else sys.stdout
    out.write(",".join(want) + ",dur_ns\n")
    i_s, i_e = want.index("start"), want.index("end")
    for r in rows:
        out.write(",".join(str(v).replace(",", ";")[:60] for v in r) + ",%d\n" % (r[i_e] - r[i_s]))


if __name__ == "__main__":
    main()
