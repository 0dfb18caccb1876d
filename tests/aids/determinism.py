#!/usr/bin/env python
"""Run the golden 64x64 batch-2 configuration twice from identical state and compare the post-step arenas bitwise (debugging aid).
usage: python tests/aids/determinism.py [mode] [steps] [overlap]"""
import contextlib
import importlib
import io
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
PKG = "semi-supervised-segmentation-cyclegan_amd"
F = importlib.import_module(PKG + ".functional")
md = importlib.import_module(PKG + ".model")
from oracle import fixtures as FX  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "f32s"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
overlap = (sys.argv[3] != "0") if len(sys.argv) > 3 else True
C, H, B = 21, 64, 2
dev = torch.device("cuda:0")
F.set_conv_precision(mode)


def run():
    args = FX.make_args(dataset="voc2012", crop_height=H, crop_width=H, batch_size=B, gpu_ids=[0], checkpoint_dir="/tmp/sscg_det", as_written=True)
    args.overlap_d = overlap
    with contextlib.redirect_stdout(io.StringIO()):
        m = md.semisuper_cycleGAN(args)
    for k, sd in FX.semisup_state_dicts(C, torch.float32, "dp").items():
        getattr(m, k).load_state_dict(sd, strict=True)
    np.random.seed(0)
    snaps = []
    for s in range(steps):
        l_img, l_gt, unl_img = FX.step_batch("dp/r0", s, C, H, H, B)
        out = m.step(l_img.to(dev), l_gt.to(dev), unl_img.to(dev))
        m.sync_losses()
        torch.cuda.synchronize()
        snaps.append((m.g_optimizer.grad.clone(), m.g_optimizer.arena.clone(), m.d_optimizer.arena.clone(), {k: float(v) for k, v in out.items()}))
    return snaps


a, b = run(), run()
for s, (x, y) in enumerate(zip(a, b)):
    print("step %d: g grad bitwise %s (max diff %.3e), g arena bitwise %s, d arena bitwise %s, losses equal %s" % (
        s, torch.equal(x[0], y[0]), float((x[0] - y[0]).abs().max()), torch.equal(x[1], y[1]), torch.equal(x[2], y[2]), x[3] == y[3]))
