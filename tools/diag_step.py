#!/usr/bin/env python
"""Step-by-step health check of a bench configuration: losses and non-finite counts of the gradient / weight arenas per step.
usage: python tools/diag_step.py --config 3 --batch 16 --steps 6 [--dtype bf16]"""
import argparse
import contextlib
import importlib
import io
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

PKG = bench.PKG
ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=3)
ap.add_argument("--batch", type=int, default=None)
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--dtype", default=None)
ap.add_argument("--no-overlap", action="store_true")
ap.add_argument("--no-sync", action="store_true", help="no host synchronisation between steps (the bench's schedule)")
ap.add_argument("--no-fork", action="store_true")
ap.add_argument("--no-side", action="store_true", help="everything on one stream")
ap.add_argument("--no-stack", action="store_true")
ap.add_argument("--time-only", action="store_true", help="with --no-sync: no per-step health record (it orders the main stream behind the D stream)")
a = ap.parse_args()
cfg = bench.CONFIGS[a.config]
dtype = a.dtype or cfg["dtype"]
bsz = a.batch or cfg["B"]
md = importlib.import_module(PKG + ".model")
F = importlib.import_module(PKG + ".functional")
data = importlib.import_module(PKG + ".data")
import main as cli  # noqa: E402
args = cli.get_args(["--model", "semisupervised_cycleGAN", "--dataset", cfg["dataset"], "--crop_height", str(cfg["H"]), "--crop_width", str(cfg["W"]),
                     "--batch_size", str(bsz), "--checkpoint_dir", "/tmp/sscg_diag_ckpt", "--dtype", dtype])
args.gpu_ids, args.as_written, args.overlap_d = [0], True, not a.no_overlap
args.fork_forward = not a.no_fork
args.stack_gsi = not a.no_stack
torch.cuda.set_device(0)
torch.manual_seed(0)
F.set_conv_precision(dtype)
if a.no_side:
    F.SideStream.enabled = False
with contextlib.redirect_stdout(io.StringIO()):
    m = md.semisuper_cycleGAN(args)
dev = torch.device("cuda", 0)
lab = list(data.SyntheticLoader(bsz, cfg["C"], cfg["H"], cfg["W"], a.steps, 1, device=dev))
unl = list(data.SyntheticLoader(bsz, cfg["C"], cfg["H"], cfg["W"], a.steps, 2, device=dev))


def bad(t):
    return int((~torch.isfinite(t)).sum())


import time  # noqa: E402
trace = []
t_start = None
host_s = 0.0
for s in range(a.steps):
    if s == 2:
        torch.cuda.synchronize()
        t_start = time.perf_counter()
    t_h = time.perf_counter()
    out = m.step(lab[s][0], lab[s][1], unl[s][0])
    if s >= 2:
        host_s += time.perf_counter() - t_h
    if a.no_sync and not a.time_only:      # device-side health record, no host synchronisation: [9 losses, non-finite counts of G.grad, D.grad, G.w]
        cur = torch.cuda.current_stream(dev)
        if args.overlap_d:
            cur.wait_stream(F.d_stream(dev))
        rec = torch.stack([v.float() for v in out.values()] + [(~torch.isfinite(t)).sum().float() for t in
                          (m.g_optimizer.grad, m.d_optimizer.grad, m.g_optimizer.arena)])
        trace.append(rec)
    if a.no_sync and s + 1 < a.steps:
        continue
    m.sync_losses()
    torch.cuda.synchronize()
    vals = {k: float(v) for k, v in out.items()}
    print("step %d  " % s + "  ".join("%s=%.4g" % (k[:14], v) for k, v in vals.items()))
    print("        non-finite: G.grad %d  D.grad %d  G.w %d  D.w %d  | max|G.grad| %.3g  max|G.w| %.3g  max|D.w| %.3g" % (
        bad(m.g_optimizer.grad), bad(m.d_optimizer.grad), bad(m.g_optimizer.arena), bad(m.d_optimizer.arena),
        float(m.g_optimizer.grad.abs().max()), float(m.g_optimizer.arena.abs().max()), float(m.d_optimizer.arena.abs().max())))
    sys.stdout.flush()
    if not all(v == v and abs(v) != float("inf") for v in vals.values()) or bad(m.g_optimizer.grad) or bad(m.d_optimizer.grad):
        for name in ("Gis", "Gsi", "Di", "Ds"):
            net = getattr(m, name)
            gbad = [k for k, p in net.named_parameters() if getattr(p, "_sscg_grad", None) is not None and bad(p._sscg_grad)]
            wbad = [k for k, p in net.named_parameters() if bad(p.data)]
            bbad = [k for k, b in net.named_buffers() if b.dtype.is_floating_point and bad(b)]
            print("   %s: %d params with non-finite grad %s | %d non-finite weights %s | %d non-finite buffers %s" % (
                name, len(gbad), gbad[:4], len(wbad), wbad[:3], len(bbad), bbad[:4]))
        break
if trace:
    torch.cuda.synchronize()
    for i, r in enumerate(trace):
        v = r.cpu().tolist()
        print("trace step %d: losses finite %s  G.grad bad %d  D.grad bad %d  G.w bad %d  | %s" % (
            i, all(x == x and abs(x) != float("inf") for x in v[:9]), v[9], v[10], v[11], " ".join("%.3g" % x for x in v[:9])))
if t_start is not None and a.no_sync:
    torch.cuda.synchronize()
    print("%.2f ms/step over steps 2..%d; host time inside step(): %.2f ms/step" % (
        (time.perf_counter() - t_start) * 1e3 / (a.steps - 2), a.steps - 1, host_s * 1e3 / (a.steps - 2)))
print("peak memory %.1f GB" % (torch.cuda.max_memory_allocated() / 2 ** 30))
