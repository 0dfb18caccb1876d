#!/usr/bin/env python
"""Split-contraction conv kernels (conv_split.hip) on the MI355X: error against fp64 beside the exact-fp32 kernel, and per-shape
timing of every tile class (tuning aid; not part of the product or the tests).
usage: python tools/convs_bench.py [check|time] [tuning codes ...]"""
import importlib
import os
import sys

import torch
import torch.nn.functional as TF

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
F = importlib.import_module("semi-supervised-segmentation-cyclegan_amd.functional")
dev = torch.device("cuda:0")
CL = torch.channels_last

# N, C, H, W, K, R, stride, pad, dil
SHAPES = [
    (8, 256, 33, 33, 256, 3, 1, 2, 2),
    (16, 256, 33, 33, 256, 3, 1, 2, 2),
    (8, 512, 33, 33, 512, 3, 1, 4, 4),
    (8, 256, 33, 33, 1024, 1, 1, 0, 1),
    (8, 1024, 33, 33, 256, 1, 1, 0, 1),
    (16, 256, 33, 33, 1024, 1, 1, 0, 1),
    (16, 1024, 33, 33, 256, 1, 1, 0, 1),
    (8, 512, 33, 33, 2048, 1, 1, 0, 1),
    (16, 256, 64, 64, 256, 3, 1, 1, 1),
    (8, 64, 65, 65, 64, 3, 1, 1, 1),
    (8, 128, 33, 33, 128, 3, 1, 1, 1),
    (8, 64, 256, 256, 128, 1, 1, 0, 1),
    (16, 64, 256, 256, 128, 3, 2, 1, 1),
    (16, 128, 128, 128, 256, 3, 2, 1, 1),
    (16, 512, 33, 33, 512, 3, 1, 4, 4),
    (16, 512, 33, 33, 2048, 1, 1, 0, 1),
    (8, 2048, 33, 33, 512, 1, 1, 0, 1),
    (8, 1024, 33, 33, 2048, 1, 1, 0, 1),
    (8, 64, 65, 65, 256, 1, 1, 0, 1),
    (8, 256, 65, 65, 64, 1, 1, 0, 1),
    (8, 128, 33, 33, 512, 1, 1, 0, 1),
    (8, 512, 33, 33, 128, 1, 1, 0, 1),
    (8, 512, 33, 33, 1024, 1, 1, 0, 1),
    (8, 1024, 33, 33, 512, 1, 1, 0, 1),
    (8, 64, 256, 256, 64, 3, 1, 1, 1),
]
CHECK = [(2, 64, 32, 32, 128, 3, 2, 1, 1), (2, 128, 16, 16, 256, 3, 2, 1, 1), (2, 256, 9, 9, 256, 3, 1, 2, 2), (3, 64, 17, 17, 64, 3, 1, 1, 1),
         (2, 256, 33, 33, 1024, 1, 1, 0, 1), (8, 256, 33, 33, 256, 3, 1, 2, 2), (2, 64, 16, 16, 128, 4, 2, 1, 1), (1, 96, 20, 20, 160, 3, 1, 1, 1)]


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


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max())


def check(tunings):
    for case in CHECK:
        n, c, h, w, k, r, s, p, d = case
        g = torch.Generator().manual_seed(1)
        x = torch.randn(n, c, h, w, generator=g)
        wt = torch.randn(k, c, r, r, generator=g) * (1.0 / (c * r * r) ** 0.5)
        xr, wr = x.double().requires_grad_(True), wt.double().requires_grad_(True)
        yr = TF.conv2d(xr, wr, None, s, p, d)
        gy = torch.randn(yr.shape, generator=g)
        yr.backward(gy.double())
        xg = x.to(dev).contiguous(memory_format=CL)
        wg = wt.to(dev).contiguous(memory_format=CL)
        gyg = gy.to(dev).contiguous(memory_format=CL)
        row = "%-36s" % (case,)
        for mode, tun in [("f32", 0)] + [("f32s", t) for t in tunings]:
            F.set_conv_precision(mode)
            F.TUNING[0] = tun
            y = F.conv2d_fwd(xg, wg, None, s, p, d)
            wtt = F.weight_transposed(wg, "x3" if (mode == "f32s" and F.split_applies(x.shape, wt.shape, s, p, d, 0, 1)) else torch.float32)
            dx = F.conv2d_dgrad(gyg, wtt, x.shape, wt.shape, s, p, d)
            row += " | %s/%x fwd %.1e dgrad %.1e" % (mode, tun, rel(y, yr), rel(dx, xr.grad))
        F.set_conv_precision("f32")
        F.TUNING[0] = 0
        print(row)
        sys.stdout.flush()


def bench(tunings):
    idx = [int(i) for i in os.environ.get("SHAPE_IDX", "").split(",") if i]
    for (N, C, H, W, K, R, s, p, d) in ([SHAPES[i] for i in idx] if idx else SHAPES):
        x = torch.randn(N, C, H, W, device=dev).contiguous(memory_format=CL)
        w = (torch.randn(K, C, R, R, device=dev) * 0.05).contiguous(memory_format=CL)
        F.set_conv_precision("f32x")
        y = F.conv2d_fwd(x, w, None, s, p, d)
        wt = F.weight_transposed(w)
        gy = torch.randn_like(y)
        flops = 2.0 * y.numel() * C * R * R
        line = "%-40s" % ("%dx%dx%d c%d k%d r%d s%d d%d" % (N, H, W, C, K, R, s, d))
        tf = timeit(lambda: F.conv2d_fwd(x, w, None, s, p, d))
        td = timeit(lambda: F.conv2d_dgrad(gy, wt, x.shape, w.shape, s, p, d))
        line += " | exact f %5.1f d %5.1f" % (flops / tf / 1e12, flops / td / 1e12)
        F.set_conv_precision("f32s")
        wt3 = F.weight_transposed(w, "x3")
        for tun in tunings:
            F.TUNING[0] = tun
            try:
                tf = timeit(lambda: F.conv2d_fwd(x, w, None, s, p, d))
                td = timeit(lambda: F.conv2d_dgrad(gy, wt3, x.shape, w.shape, s, p, d))
                line += " | %3x f %5.1f d %5.1f" % (tun, flops / tf / 1e12, flops / td / 1e12)
            except Exception as e:
                line += " | %3x ERR %s" % (tun, str(e)[:30])
        F.TUNING[0] = 0
        F.set_conv_precision("f32")
        print(line)
        sys.stdout.flush()


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "check"
    tun = [int(a, 0) for a in sys.argv[2:]] or [0]
    (check if what == "check" else bench)(tun)
