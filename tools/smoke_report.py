#!/usr/bin/env python
"""The smoke step's nine first-step losses against the CPU oracle (fp32 and fp64), printed, nothing asserted: how far a change of
arithmetic moves the chained losses of ONE seed.  Run under SSCG_LIB / SSCG_FUSE_HEAD / ... variants (tools/r4_probe28.sh)."""
import contextlib
import io
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G      # noqa: E402
from oracle import fixtures as FX   # noqa: E402
from oracle import step as ostep    # noqa: E402

dev = torch.device("cuda:0")
C, H, B = 21, 64, 2
md = G.load("model")
F = G.load("functional")
args = FX.make_args(dataset="voc2012", crop_height=H, crop_width=H, batch_size=B, gpu_ids=[0], checkpoint_dir="/tmp/sscg_smoke_ckpt", as_written=True)
with contextlib.redirect_stdout(io.StringIO()):
    m = md.semisuper_cycleGAN(args)
sds = FX.semisup_state_dicts(C, torch.float32, "smoke")
l_img, l_gt, unl_img = FX.step_batch("smoke", 0, C, H, H, B)
np.random.seed(0)
ref = ostep.SemiSupOracle(C, FX.semisup_state_dicts(C, torch.float32, "smoke"), crop=(H, H)).step(l_img, l_gt, unl_img)
np.random.seed(0)
r64 = ostep.SemiSupOracle(C, FX.semisup_state_dicts(C, torch.float64, "smoke"), crop=(H, H)).step(l_img.double(), l_gt, unl_img.double())
tag = " ".join("%s=%s" % (k, os.path.basename(v)) for k, v in sorted(os.environ.items()) if k.startswith("SSCG_"))
for mode in ("f32s", "f32x"):
    F.set_conv_precision(mode)
    for k, sd in sds.items():
        getattr(m, k).load_state_dict(sd, strict=True)
    np.random.seed(0)
    out = m.step(l_img.to(dev), l_gt.to(dev), unl_img.to(dev))
    m.sync_losses()
    got = {k: float(v) for k, v in out.items()}
    print("[%s] %s: " % (tag or "default", mode) + "  ".join("%s %+.2e" % (k.replace("_loss", ""), (got[k] - r64[k]) / abs(r64[k])) for k in FX.CHAINED_LOSSES) +
          "  | worst one-pass-deep %.1e" % max(abs(got[k] - r64[k]) / abs(r64[k]) for k in ostep.LOSS_KEYS if k not in FX.CHAINED_LOSSES))
print("[oracle fp32 vs fp64] " + "  ".join("%s %+.2e" % (k.replace("_loss", ""), (ref[k] - r64[k]) / abs(r64[k])) for k in FX.CHAINED_LOSSES))
