#!/usr/bin/env python
"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel name (mean per launch) into one small JSON.
usage: python tools/pmc_summary.py OUT.json DIR [DIR ...]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    name = row.get("Kernel_Name") or row.get("Kernel Name") or "?"
                    cname = row.get("Counter_Name") or row.get("Counter Name")
                    val = float(row.get("Counter_Value") or row.get("Counter Value") or 0)
                    a = agg[name][cname]
                    a[0] += 1
                    a[1] += val
    res = {}
    for name, cs in agg.items():
        res[name] = {c: {"launches": n, "mean": s / max(n, 1), "sum": s} for c, (n, s) in cs.items()}
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1, sort_keys=True)
    # console digest: conv kernels
    for name, cs in sorted(res.items(), key=lambda kv: -sum(v["sum"] for v in kv[1].values())):
        if "conv_" in name or "norm" in name or "col_reduce" in name:
            print(name[:90], {c: "%.3g/launch x%d" % (v["mean"], v["launches"]) for c, v in cs.items()})


if __name__ == "__main__":
    main()
