#!/usr/bin/env python
"""Error of the three conv products against fp64 in the exact (f32) and split-bf16 (f32s) modes on the ResnetGenerator's layer shapes."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn.functional as TF  # noqa: E402

F = importlib.import_module("semi-supervised-segmentation-cyclegan_amd.functional")
dev = torch.device("cuda", 0)
CL = torch.channels_last


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max())


CASES = [(2, 64, 32, 32, 128, 3, 2, 1, 1), (2, 128, 16, 16, 256, 3, 2, 1, 1), (2, 256, 8, 8, 256, 3, 1, 1, 1), (2, 21, 38, 38, 64, 7, 1, 0, 1),
         (2, 64, 38, 38, 3, 7, 1, 0, 1), (2, 64, 16, 16, 128, 4, 2, 1, 1)]
for case in CASES:
    n, c, h, w, k, r, s, p, d = case
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, c, h, w, generator=g)
    wt = torch.randn(k, c, r, r, generator=g) * (1.0 / (c * r * r) ** 0.5)
    xr, wr = x.double().requires_grad_(True), wt.double().requires_grad_(True)
    yr = TF.conv2d(xr, wr, None, s, p, d)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy.double())
    row = "%-34s" % (case,)
    for mode in ("f32", "f32s"):
        F.set_conv_precision(mode)
        xg = x.to(dev).contiguous(memory_format=CL)
        wg = wt.to(dev).contiguous(memory_format=CL)
        gyg = gy.to(dev).contiguous(memory_format=CL)
        y = F.conv2d_fwd(xg, wg, None, s, p, d)
        dx = F.conv2d_dgrad(gyg, F.weight_transposed(wg), x.shape, wt.shape, s, p, d)
        dw = F.conv2d_wgrad(xg, gyg, wt.shape, s, p, d)
        row += " | %s fwd %.1e dgrad %.1e wgrad %.1e" % (mode, rel(y, yr), rel(dx, xr.grad), rel(dw, wr.grad))
    F.set_conv_precision("f32")
    print(row)
