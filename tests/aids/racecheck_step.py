#!/usr/bin/env python
"""Run a few training steps under the stream-ordering checker (racecheck.py) and print what it found.
usage: [SSCG_SIDE_LANES=3] [SSCG_FORCE_DP=1] python tests/aids/racecheck_step.py [steps] [size] [batch] [overlap 0|1] [dtype]
No host synchronisation between the steps (as bench.py / main.py run them): step N+1 is issued while step N's D step is in flight."""
import contextlib
import importlib
import io
import os
import sys

os.environ["SSCG_RACECHECK"] = "1"
import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import PKG_NAME  # noqa: E402
from oracle import fixtures as FX  # noqa: E402

md = importlib.import_module(PKG_NAME + ".model")
F = importlib.import_module(PKG_NAME + ".functional")
rc = importlib.import_module(PKG_NAME + "._lib").dev_tool("racecheck")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
size = int(sys.argv[2]) if len(sys.argv) > 2 else 64
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 2
overlap = (sys.argv[4] != "0") if len(sys.argv) > 4 else True
dtype = sys.argv[5] if len(sys.argv) > 5 else "f32"
dev = torch.device("cuda", 0)
dp = None
if os.environ.get("SSCG_FORCE_DP"):
    dp = importlib.import_module(PKG_NAME + ".parallel").DataParallel()
F.set_conv_precision(dtype)
args = FX.make_args(dataset="voc2012", crop_height=size, crop_width=size, batch_size=batch, gpu_ids=[0], checkpoint_dir="/tmp/sscg_rc", as_written=True)
args.overlap_d = overlap
with contextlib.redirect_stdout(io.StringIO()):
    m = md.semisuper_cycleGAN(args, data_parallel=dp)
np.random.seed(0)
batches = [tuple(t.to(dev) for t in FX.step_batch("pool", s % 4, 21, size, size, batch)) for s in range(4)]
torch.cuda.synchronize()
for s in range(steps):
    l_img, l_gt, unl_img = batches[s % 4]
    out = m.step(l_img, l_gt, unl_img)
    rc.name_streams(F, dev)
    print("step %d issued: %d launches, %d reports so far" % (s, rc.CORE.launches, len(rc.CORE.reports)), flush=True)
m.sync_losses()
torch.cuda.synchronize()
print("losses finite:", all(bool(torch.isfinite(v)) for v in out.values()))
core = rc.report(sys.stdout)
sys.exit(1 if core.reports else 0)
