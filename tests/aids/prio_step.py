#!/usr/bin/env python
"""Two training steps at 64x64, batch 2 with whatever side-lane priority the environment selects (SSCG_SIDE_PRIORITY); prints the
losses and a checksum of the generator arena as one JSON line (tests/test_step_gpu.py compares the two priorities bit for bit)."""
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
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import PKG_NAME  # noqa: E402
from oracle import fixtures as FX  # noqa: E402

md = importlib.import_module(PKG_NAME + ".model")
F = importlib.import_module(PKG_NAME + ".functional")
dev = torch.device("cuda", 0)
args = FX.make_args(dataset="voc2012", crop_height=64, crop_width=64, batch_size=2, gpu_ids=[0], checkpoint_dir="/tmp/sscg_prio", as_written=True)
args.overlap_d = True
with contextlib.redirect_stdout(io.StringIO()):
    m = md.semisuper_cycleGAN(args)
for k, sd in FX.semisup_state_dicts(21, torch.float32, "prio").items():
    getattr(m, k).load_state_dict(sd, strict=True)
np.random.seed(0)
out = None
for s in range(2):
    out = m.step(*[t.to(dev) for t in FX.step_batch("prio", s, 21, 64, 64, 2)])
m.sync_losses()
torch.cuda.synchronize()
g = m.g_optimizer.arena.detach().double()
print(json.dumps({"priority": F.SideStream.priority, "losses": {k: float(v).hex() for k, v in out.items()},
                  "g_sum": float(g.sum()).hex(), "g_abs": float(g.abs().sum()).hex(), "d_sum": float(m.d_optimizer.arena.detach().double().sum()).hex()}))
