#!/usr/bin/env python
"""Repeat the pool-swap run of tests/test_parity_gpu.py (64 steps at 32x32, batch 2, D step overlapped) and report the first
non-finite loss, if any.  usage: python tests/aids/flake_pool.py [runs] [overlap 0|1]"""
import contextlib
import importlib
import io
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import PKG_NAME  # noqa: E402
from oracle import fixtures as FX  # noqa: E402

md = importlib.import_module(PKG_NAME + ".model")
F = importlib.import_module(PKG_NAME + ".functional")
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 10
overlap = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
dev = torch.device("cuda", 0)
bad = 0
for it in range(runs):
    args = FX.make_args(dataset="voc2012", crop_height=32, crop_width=32, batch_size=2, gpu_ids=[0], checkpoint_dir="/tmp/sscg_flake", as_written=True)
    args.overlap_d = overlap
    with contextlib.redirect_stdout(io.StringIO()):
        m = md.semisuper_cycleGAN(args)
    for k, sd in FX.semisup_state_dicts(21, torch.float32, "pool").items():
        getattr(m, k).load_state_dict(sd, strict=True)
    np.random.seed(0)
    first = None
    for s in range(64):
        l_img, l_gt, unl_img = FX.step_batch("pool", s % 4, 21, 32, 32, 2)
        out = m.step(l_img.to(dev), l_gt.to(dev), unl_img.to(dev))
        if s % 8 != 7:              # (the host reads the losses every eighth step only: in between, step N+1 is issued under D step N)
            continue
        m.sync_losses()
        vals = {k: float(v) for k, v in out.items()}
        if first is None and not all(np.isfinite(v) for v in vals.values()):
            first = (s, {k: v for k, v in vals.items() if not np.isfinite(v)})
            break
    torch.cuda.synchronize()
    if first is not None:
        bad += 1
        nan_g = bool(torch.isnan(m.g_optimizer.arena).any())
        nan_d = bool(torch.isnan(m.d_optimizer.arena).any())
        print("run %d: first non-finite at step %d: %s; NaN in G arena %s, D arena %s" % (it, first[0], first[1], nan_g, nan_d), flush=True)
    del m
print("%d of %d runs non-finite" % (bad, runs))
