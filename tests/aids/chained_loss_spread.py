#!/usr/bin/env python
"""Spread of the first-step losses of the 256x256 golden configuration over equally valid arithmetic variants of the build (exact /
split contraction, forced tile classes): how much of a chained loss's distance to the fp64 oracle is summation-order noise
(SURVEY App. D).  Debugging aid.  usage: python tests/aids/chained_loss_spread.py [config]"""
import contextlib
import importlib
import io
import json
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
from oracle import step as ostep  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "s256"
meta = json.load(open(os.path.join(ROOT, "tests", "golden", "meta.json")))
info = meta["g3"][cfg]
C, dataset, H, Wd, B, steps = FX.STEP_CONFIGS[cfg]
dev = torch.device("cuda:0")
ref32, ref64 = info["reference_f32"][0], info["oracle_f64"][0]
KEYS = ("img_cycle_loss", "gt_cycle_loss", "cycle_img_dis_loss", "lab_loss_CE", "img_gen_loss")
print("reference fp32 vs fp64: " + "  ".join("%s %.2e" % (k, abs(ref32[k] - ref64[k]) / abs(ref64[k])) for k in KEYS))
for mode, tun in (("f32x", (None, 0)), ("f32x", (3, 0)), ("f32x", (6, 0)), ("f32x", (7, 0)), ("f32s", (None, 0)), ("f32s", (0, 0)), ("f32s", (2, 0)), ("f32s", (4, 0)),
                  ("f32s", (None, 1))):
    F.set_conv_precision(mode)
    F.tuning(tile_class=tun[0], split=tun[1])
    args = FX.make_args(dataset=dataset, crop_height=H, crop_width=Wd, batch_size=B, gpu_ids=[0], checkpoint_dir="/tmp/sscg_spread", as_written=True)
    with contextlib.redirect_stdout(io.StringIO()):
        m = md.semisuper_cycleGAN(args)
    for k, sd in FX.semisup_state_dicts(C, torch.float32, cfg).items():
        getattr(m, k).load_state_dict(sd, strict=True)
    np.random.seed(0)
    l_img, l_gt, unl_img = FX.step_batch(cfg, 0, C, H, Wd, B)
    try:
        got = {k: float(v) for k, v in m.step(l_img.to(dev), l_gt.to(dev), unl_img.to(dev)).items()}
        print("%-5s tile %-5s split %d: " % (mode, tun[0], tun[1]) + "  ".join("%s %+.2e" % (k, (got[k] - ref64[k]) / abs(ref64[k])) for k in KEYS))
    except Exception as e:
        print("%-5s tile %-5s split %d: ERR %s" % (mode, tun[0], tun[1], str(e)[:80]))
    F.tuning()
    del m
