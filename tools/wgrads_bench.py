#!/usr/bin/env python
"""Weight-gradient kernels on the MI355X: exact fp32 MFMA, on-the-fly split (conv_wgrad.hip) and the planes-based split kernel
(conv_split.hip wgrads_kernel), error against fp64 samples and TFLOP/s per shape (tuning aid).  usage: python tools/wgrads_bench.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
F = importlib.import_module("semi-supervised-segmentation-cyclegan_amd.functional")
dev = torch.device("cuda:0")
CL = torch.channels_last
SHAPES = [(8, 256, 33, 33, 256, 3, 1, 2, 2), (16, 256, 33, 33, 256, 3, 1, 2, 2), (8, 512, 33, 33, 512, 3, 1, 4, 4), (8, 256, 33, 33, 1024, 1, 1, 0, 1),
          (8, 1024, 33, 33, 256, 1, 1, 0, 1), (8, 512, 33, 33, 2048, 1, 1, 0, 1), (8, 128, 33, 33, 128, 3, 1, 1, 1), (8, 64, 256, 256, 128, 1, 1, 0, 1),
          (16, 128, 128, 128, 256, 3, 2, 1, 1), (3, 136, 19, 23, 200, 3, 1, 1, 1)]


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


for (N, C, H, W, K, R, s, p, d) in SHAPES:
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(N, C, H, W, device=dev, generator=g).contiguous(memory_format=CL)
    P, Q = F.conv_out_size(H, R, s, p, d), F.conv_out_size(W, R, s, p, d)
    dy = torch.randn(N, K, P, Q, device=dev, generator=g).contiguous(memory_format=CL)
    flops = 2.0 * N * P * Q * K * C * R * R
    F.set_conv_precision("f32x")
    ref = F.conv2d_wgrad(x, dy, (K, C, R, R), s, p, d)
    line = "%-40s" % ("%dx%dx%d c%d k%d r%d s%d d%d" % (N, H, W, C, K, R, s, d))
    line += " | exact %5.1f" % (flops / timeit(lambda: F.conv2d_wgrad(x, dy, (K, C, R, R), s, p, d)) / 1e12)
    F.set_conv_precision("f32s")
    for name, kw in (("fly", dict(wgrad_class=2)), ("planes", dict()), ("planes/2st", dict(wgrad_flags=1))):
        old = F.tuning(**kw)
        try:
            dw = F.conv2d_wgrad(x, dy, (K, C, R, R), s, p, d)
            err = float((dw - ref).abs().max() / ref.abs().max())
            line += " | %s %5.1f (vs exact %.1e)" % (name, flops / timeit(lambda: F.conv2d_wgrad(x, dy, (K, C, R, R), s, p, d)) / 1e12, err)
        except Exception as e:
            line += " | %s ERR %s" % (name, str(e)[:40])
        F.TUNING[0], F.WGRAD_TUNING[0] = old
    F.set_conv_precision("f32")
    print(line)
    sys.stdout.flush()
