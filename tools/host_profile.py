#!/usr/bin/env python
"""Where the host thread's time of one step goes: cProfile over dry-run steps (every library launch a no-op, sscg_set_dry_run) of the
64x64 batch-2 case - the host-bound case of bench.py.  usage: python tools/host_profile.py [steps] [size] [batch] > profile.txt"""
import contextlib
import cProfile
import importlib
import io
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import PKG_NAME  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
size = int(sys.argv[2]) if len(sys.argv) > 2 else 64
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 2
md = importlib.import_module(PKG_NAME + ".model")
import main as cli  # noqa: E402  (the product CLI: its defaults)
data = importlib.import_module(PKG_NAME + ".data")
lib = importlib.import_module(PKG_NAME + "._lib").lib
dev = torch.device("cuda", 0)
args = cli.get_args(["--model", "semisupervised_cycleGAN", "--dataset", "voc2012", "--crop_height", str(size), "--crop_width", str(size),
                     "--batch_size", str(batch), "--checkpoint_dir", "/tmp/sscg_hostprof", "--dtype", "f32"])
args.gpu_ids, args.as_written = [0], True
with contextlib.redirect_stdout(io.StringIO()):
    m = md.semisuper_cycleGAN(args)
n = steps + 6
sl = list(data.SyntheticLoader(batch, 21, size, size, n, 3, device=dev))
su = list(data.SyntheticLoader(batch, 21, size, size, n, 4, device=dev))
for i in range(4):
    m.step(sl[i][0], sl[i][1], su[i][0])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(4, 6):
    m.step(sl[i][0], sl[i][1], su[i][0])
th = (time.perf_counter() - t0) / 2
torch.cuda.synchronize()
print("real steps: host issue %.1f ms, step %.1f ms" % (1e3 * th, 1e3 * (time.perf_counter() - t0) / 2))
lib.sscg_set_dry_run(1)
m.step(sl[5][0], sl[5][1], su[5][0])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(6, 6 + steps):
    m.step(sl[i][0], sl[i][1], su[i][0])
print("dry steps, no profiler: %.1f ms per step" % (1e3 * (time.perf_counter() - t0) / steps))
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(6, 6 + steps):
    m.step(sl[i][0], sl[i][1], su[i][0])
pr.disable()
torch.cuda.synchronize()
lib.sscg_set_dry_run(0)
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    print(s.getvalue().replace(ROOT + "/", ""))
