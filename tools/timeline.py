#!/usr/bin/env python
"""Stream-level timeline of a rocprofv3 --kernel-trace run of bench.py: for the last `steps` training steps (delimited by the optimiser
launches) - per stream: kernels, busy time, the gaps BETWEEN consecutive kernels of that stream (dependency / issue stalls of a lane);
over all streams: how long exactly k kernels were in flight; and the kernels that ran ALONE for the longest total time (what the chip
executes with nothing beside it).  usage: python tools/timeline.py <results.db> [steps]"""
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    i = name.find("(")
    return (name[:i] if i > 0 else name)[:60]


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    rows = db.execute("select start, end, stream_id, name from kernels order by start").fetchall()
    adam = [r[1] for r in rows if "adam_kernel" in r[3]]
    lo, hi = adam[-1 - 2 * steps], adam[-1]
    rows = [r for r in rows if r[0] >= lo and r[1] <= hi]
    span = hi - lo
    print("window: %d steps, %.2f ms per step, %d kernels per step" % (steps, span / steps / 1e6, len(rows) // steps))
    per = defaultdict(list)
    for s, e, sid, n in rows:
        per[sid].append((s, e, n))
    print("per stream (ms per step): kernels | busy | gaps between its kernels: <3us, 3-10us, 10-30us, 30-100us, >100us (count / ms)")
    for sid, ks in sorted(per.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1])):
        busy = sum(e - s for s, e, _ in ks)
        bins = [[0, 0] for _ in range(5)]
        for (s0, e0, _), (s1, e1, _) in zip(ks, ks[1:]):
            g = max(0, s1 - e0)
            b = 0 if g < 3000 else 1 if g < 10000 else 2 if g < 30000 else 3 if g < 100000 else 4
            bins[b][0] += 1
            bins[b][1] += g
        print("  stream %-4s %5d | %6.1f | %s" % (sid, len(ks) // steps, busy / steps / 1e6,
                                                 "  ".join("%d / %.1f" % (c // steps, t / steps / 1e6) for c, t in bins)))
    # what the generator's optimiser launch waits for: per stream, when its last kernel before each adam launch of the window ended
    adam_rows = [r for r in rows if "adam_kernel" in r[3]]
    big = [r for r in adam_rows if (r[1] - r[0]) > 100000]          # the generator arena's launch (85.7 M elements: hundreds of us)
    for a in big[:steps]:
        t0 = a[0]
        tails = {}
        for sid, ks in per.items():
            prev = [k for k in ks if k[1] <= t0]
            if prev:
                tails[sid] = (t0 - prev[-1][1], short(prev[-1][2]))
        print("before the generator update at +%.1f ms: " % ((t0 - lo) / 1e6) + "; ".join("stream %s idle for %.0f us (last: %s)" % (sid, g / 1e3, n[:28]) for sid, (g, n) in sorted(tails.items())))
    # concurrency histogram by sweeping events
    ev = []
    for i, (s, e, sid, n) in enumerate(rows):
        ev.append((s, 1, i))
        ev.append((e, -1, i))
    ev.sort()
    active = set()
    hist = defaultdict(int)
    alone = defaultdict(int)
    prev = ev[0][0]
    for t, d, i in ev:
        k = len(active)
        hist[k] += t - prev
        if k == 1:
            alone[short(rows[next(iter(active))][3])] += t - prev
        prev = t
        if d > 0:
            active.add(i)
        else:
            active.discard(i)
    print("kernels in flight (ms per step): " + "  ".join("%d: %.1f" % (k, v / steps / 1e6) for k, v in sorted(hist.items())))
    print("running ALONE (ms per step), top 12:")
    for n, v in sorted(alone.items(), key=lambda kv: -kv[1])[:12]:
        print("  %6.2f  %s" % (v / steps / 1e6, n))


    gantt(rows, per, lo + (steps - 1) * span // steps, hi)


def gantt(rows, per, t0, t1, bucket_us=1000):
    """The last step as a chart: one row per stream, one character per bucket - the busy fraction of the stream in that bucket
    (' ' idle, '.' < 1/3, ':' < 2/3, '#' more) - and under it the kind of kernel that filled most of the bucket
    (F / D forward conv / data gradient of the 128x64 class, f / d of the 64x64 class, R / r of the 128x128 class, k exact-fp32 stems,
    W weight gradient, n norm apply, b norm backward, o other)."""
    nb = int((t1 - t0) / (bucket_us * 1000)) + 1

    def kind(n):
        if "conv_kc_kernel" in n or "thin_" in n:
            return "k"                  # exact-fp32 stems / thin 1x1
        if "convs_kernel<0, 2, 2, 2, 2" in n:
            return "R"                  # 128x128 class forward (the ResNet generators' 64x64 maps)
        if "convs_kernel<1, 2, 2, 2, 2" in n:
            return "r"
        if "convs_kernel<0, 2, 2, 1, 1" in n:
            return "f"                  # 64x64 class
        if "convs_kernel<1, 2, 2, 1, 1" in n:
            return "d"
        if "convs_kernel<0" in n or "conv16_kernel<0" in n:
            return "F"
        if "convs_kernel<1" in n or "conv16_kernel<1" in n:
            return "D"
        if "wgrad" in n:
            return "W"
        if "norm_apply" in n:
            return "n"
        if "norm_bwd" in n:
            return "b"
        return "o"
    print("last step, %d us per character:" % bucket_us)
    for sid, ks in sorted(per.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1])):
        busy = [0.0] * nb
        kinds = [defaultdict(float) for _ in range(nb)]
        for s, e, n in ks:
            if e <= t0 or s >= t1:
                continue
            s, e = max(s, t0), min(e, t1)
            b = int((s - t0) / (bucket_us * 1000))
            while s < e:
                be = t0 + (b + 1) * bucket_us * 1000
                d = min(e, be) - s
                busy[b] += d
                kinds[b][kind(n)] += d
                s = min(e, be)
                b += 1
        frac = [x / (bucket_us * 1000) for x in busy]
        print("  stream %-3s |%s|" % (sid, "".join(" " if f < 0.02 else "." if f < 0.33 else ":" if f < 0.67 else "#" for f in frac)))
        print("             |%s|" % "".join(" " if not k else max(k.items(), key=lambda kv: kv[1])[0] for k in kinds))


if __name__ == "__main__":
    main()
