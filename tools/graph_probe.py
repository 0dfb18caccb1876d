#!/usr/bin/env python
"""Feasibility probe: capture one whole G+D step (all four lanes, both optimisers) in a HIP graph and replay it.  NOT a training
loop - the replayed step repeats the captured step's host decisions (pool slots, Adam step count, dropout seeds); what is measured
is whether the capture goes through with the ctypes launches / stream forks of the step, and what a replay costs the host and the GPU.
usage: python tools/graph_probe.py [H] [B] [dtype]"""
import contextlib
import importlib
import io
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "semi-supervised-segmentation-cyclegan_amd"
F = importlib.import_module(PKG + ".functional")
md = importlib.import_module(PKG + ".model")
data = importlib.import_module(PKG + ".data")
import main as cli  # noqa: E402

H = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dtype = sys.argv[3] if len(sys.argv) > 3 else "f32"
dev = torch.device("cuda:0")
args = cli.get_args(["--model", "semisupervised_cycleGAN", "--dataset", "voc2012", "--crop_height", str(H), "--crop_width", str(H),
                     "--batch_size", str(B), "--checkpoint_dir", "/tmp/sscg_graph_probe", "--dtype", dtype])
args.gpu_ids, args.as_written = [0], True
args.overlap_d = False            # one self-contained step per graph
F.set_conv_precision(dtype)
torch.manual_seed(0)
with contextlib.redirect_stdout(io.StringIO()):
    m = md.semisuper_cycleGAN(args)
lab = list(data.SyntheticLoader(B, 21, H, H, 4, 1, device=dev))
unl = list(data.SyntheticLoader(B, 21, H, H, 4, 2, device=dev))


def eager(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        out = m.step(lab[i % 4][0], lab[i % 4][1], unl[i % 4][0])
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    return th / n, (time.perf_counter() - t0) / n, out


for _ in range(3):
    eager(2)
th, tt, out = eager(10)
print("eager  : host issue %.2f ms/step, wall %.2f ms/step" % (th * 1e3, tt * 1e3))
# static inputs
s_img, s_gt, s_unl = lab[0][0].clone(), lab[0][1].clone(), unl[0][0].clone()
variant = sys.argv[4] if len(sys.argv) > 4 else "full"
if variant == "fwd":            # one network forward, no autograd, one stream
    def body():
        with torch.no_grad():
            return {"y": m.Gsi(s_img).sum()}
elif variant == "serial":       # the whole step on ONE stream
    F.SideStream.enabled = False
    m.fork_forward = False
    body = lambda: m.step(s_img, s_gt, s_unl)
else:
    body = lambda: m.step(s_img, s_gt, s_unl)
for _ in range(2):
    body()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    body()
th = (time.perf_counter() - t0) / 10
torch.cuda.synchronize()
print("eager %s: host issue %.2f ms, wall %.2f ms" % (variant, th * 1e3, (time.perf_counter() - t0) / 10 * 1e3))
g = torch.cuda.CUDAGraph()
try:
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        gout = body()
        F.SideStream.join(dev)          # every forked lane back into the capturing stream (the eager step leaves the operand-copy
        F.ForkStream.join(dev)          # refresh on side lane 0 for the next step to wait on)
    torch.cuda.synchronize()
except Exception as e:
    print("capture FAILED: %s: %s" % (type(e).__name__, str(e)[:300]))
    sys.exit(0)
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(10):
    g.replay()
th = (time.perf_counter() - t0) / 10
torch.cuda.synchronize()
tt = (time.perf_counter() - t0) / 10
print("replay : host issue %.2f ms/step, wall %.2f ms/step; losses finite %s" % (th * 1e3, tt * 1e3, all(bool(torch.isfinite(v)) for v in gout.values())))
