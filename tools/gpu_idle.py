#!/usr/bin/env python
"""How busy the GPU is over a rocprofv3 --kernel-trace run: union of the kernel intervals of all streams against the wall span,
per-stream busy time, and the distribution of the gaps with NO kernel in flight.  usage: python tools/gpu_idle.py <results.db> [steps]
The window is the last `steps` (default 4) training steps, delimited by the optimiser launches (two adam_kernel launches per step)."""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    rows = db.execute("select start, end, stream_id, name from kernels order by start").fetchall()
    adam = [r[1] for r in rows if "adam_kernel" in r[3]]
    lo, hi = adam[-1 - 2 * steps], adam[-1]
    rows = [r for r in rows if r[0] >= lo and r[1] <= hi]
    span = hi - lo
    print("window: %d steps, %.2f ms per step" % (steps, span / steps / 1e6))
    busy = 0
    cur_s, cur_e = rows[0][0], rows[0][1]
    gaps = []
    for s, e, _, _ in rows[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append(s - cur_e)
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    per = {}
    for s, e, sid, _ in rows:
        per[sid] = per.get(sid, 0) + (e - s)
    print("span %.1f ms, at least one kernel in flight %.1f ms (%.1f %%), sum of kernel durations %.1f ms (%.2f kernels in flight on average)" % (
        span / 1e6, busy / 1e6, 100.0 * busy / span, sum(per.values()) / 1e6, sum(per.values()) / span))
    for sid, b in sorted(per.items(), key=lambda kv: -kv[1]):
        print("  stream %s: %.1f ms of kernels (%.1f %% of the span)" % (sid, b / 1e6, 100.0 * b / span))
    gaps.sort()
    if gaps:
        tot = sum(gaps)
        print("idle gaps: %d, total %.1f ms; median %.1f us, p90 %.1f us, max %.1f us; gaps > 20 us: %d (%.1f ms)" % (
            len(gaps), tot / 1e6, gaps[len(gaps) // 2] / 1e3, gaps[int(len(gaps) * 0.9)] / 1e3, gaps[-1] / 1e3,
            sum(1 for g in gaps if g > 20000), sum(g for g in gaps if g > 20000) / 1e6))


if __name__ == "__main__":
    main()
