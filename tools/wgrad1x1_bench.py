#!/usr/bin/env python
"""1x1 weight gradients of the step's shapes: the fused-split kernel (conv_split.hip wgradf_kernel, default) beside the on-the-fly
split kernel of conv_wgrad.hip (tuning class 2) and the pre-split planes kernel (class 3); max error of a sample of outputs against
fp64.  usage: python tools/wgrad1x1_bench.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
F = importlib.import_module("semi-supervised-segmentation-cyclegan_amd.functional")
dev = torch.device("cuda:0")
CL = torch.channels_last
F.set_conv_precision("f32s")
# N, C, H, W, K
SHAPES = [(8, 256, 33, 33, 1024), (8, 1024, 33, 33, 256), (16, 256, 33, 33, 1024), (16, 1024, 33, 33, 256), (8, 512, 33, 33, 2048),
          (8, 2048, 33, 33, 512), (8, 1024, 33, 33, 2048), (8, 128, 33, 33, 512), (8, 512, 33, 33, 128), (8, 512, 33, 33, 1024),
          (8, 1024, 33, 33, 512), (8, 256, 65, 65, 128), (2, 256, 33, 33, 1024), (8, 136, 33, 33, 264)]


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


for n, c, h, w, k in SHAPES:
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn(n, c, h, w, generator=g).to(dev).contiguous(memory_format=CL)
    dy = torch.randn(n, k, h, w, generator=g).to(dev).contiguous(memory_format=CL)
    flops = 2.0 * n * h * w * c * k
    row = "%-28s" % ("%dx%dx%d c%d k%d" % (n, h, w, c, k))
    ref = None
    variants = (("fused", None, 0), ("on-the-fly", 2, 0), ("planes", 3, 0))
    if os.environ.get("WGF_SPLITS"):        # sweep of the fused kernel's pixel splits
        variants = tuple(("fused/%s" % sp, None, int(sp)) for sp in os.environ["WGF_SPLITS"].split(","))
    for name, cls, nsplit in variants:
        old = F.tuning(wgrad_class=cls, wgrad_splits=nsplit)
        try:
            dw = F.conv2d_wgrad(x, dy, (k, c, 1, 1), 1, 0, 1)
            t = timeit(lambda: F.conv2d_wgrad(x, dy, (k, c, 1, 1), 1, 0, 1))
        finally:
            F.TUNING[0], F.WGRAD_TUNING[0] = old
        if ref is None:
            # 64 sampled outputs in fp64
            ks = torch.randint(0, k, (64,), generator=g)
            cs = torch.randint(0, c, (64,), generator=g)
            xm = x.permute(0, 2, 3, 1).reshape(-1, c).double()
            ym = dy.permute(0, 2, 3, 1).reshape(-1, k).double()
            ref = (ym[:, ks.to(dev)] * xm[:, cs.to(dev)]).sum(0)
        got = dw.reshape(k, c)[ks.to(dev), cs.to(dev)].double()
        err = float((got - ref).abs().max() / ref.abs().max())
        row += " | %-10s %6.1f us %6.1f TF/s err %.1e" % (name, t * 1e6, flops / t / 1e12, err)
    print(row, flush=True)
