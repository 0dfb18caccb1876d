#!/usr/bin/env python
"""Where the host's issue time of a step goes: cProfile over steps of the 64x64 batch-2 case (the GPU is never the bound there).
usage: python tools/host_profile.py [--dtype bf16] [--steps 10] [out.txt]"""
import argparse
import contextlib
import cProfile
import importlib
import io
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--single-thread-backward", action="store_true", help="run the autograd engine on the calling thread (its Python time becomes visible)")
ap.add_argument("out", nargs="?")
a = ap.parse_args()
PKG = bench.PKG
md = importlib.import_module(PKG + ".model")
F = importlib.import_module(PKG + ".functional")
data = importlib.import_module(PKG + ".data")
import main as cli  # noqa: E402
args = cli.get_args(["--model", "semisupervised_cycleGAN", "--dataset", "cityscapes", "--crop_height", "64", "--crop_width", "64",
                     "--batch_size", "2", "--checkpoint_dir", "/tmp/sscg_hp_ckpt", "--dtype", a.dtype])
args.gpu_ids, args.as_written, args.overlap_d = [0], True, True
torch.cuda.set_device(0)
if a.single_thread_backward:
    torch.autograd.set_multithreading_enabled(False)
F.set_conv_precision(a.dtype)
with contextlib.redirect_stdout(io.StringIO()):
    m = md.semisuper_cycleGAN(args)
dev = torch.device("cuda", 0)
n = a.steps + 4
lab = list(data.SyntheticLoader(2, 20, 64, 64, n, 1, device=dev))
unl = list(data.SyntheticLoader(2, 20, 64, 64, n, 2, device=dev))
for i in range(4):
    m.step(lab[i][0], lab[i][1], unl[i][0])
torch.cuda.synchronize()
import time  # noqa: E402
t0 = time.perf_counter()
for i in range(4, n):
    m.step(lab[i][0], lab[i][1], unl[i][0])
t1 = time.perf_counter()
torch.cuda.synchronize()
print("host issue time without the profiler: %.2f ms/step" % ((t1 - t0) * 1e3 / a.steps))
pr = cProfile.Profile()
pr.enable()
for i in range(4, n):
    m.step(lab[i][0], lab[i][1], unl[i][0])
pr.disable()
torch.cuda.synchronize()
out = open(a.out, "w") if a.out else sys.stdout
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).strip_dirs().sort_stats(key).print_stats(45)
    out.write("==== sorted by %s (over %d steps) ====\n%s\n" % (key, a.steps, s.getvalue()))
