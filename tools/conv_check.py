#!/usr/bin/env python
"""Forward / data-gradient of single convolutions against torch fp64 on the MI355X (bug-localisation aid)."""
import importlib
import os
import sys

import torch
import torch.nn.functional as TF

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
F = importlib.import_module("semi-supervised-segmentation-cyclegan_amd.functional")
dev = torch.device("cuda:0")
CL = torch.channels_last

# N, C, H, W, K, R, stride, pad, dil, reflect
SHAPES = [
    (2, 3, 32, 32, 64, 7, 1, 3, 1, 1),
    (2, 64, 32, 32, 128, 3, 2, 1, 1, 0),
    (2, 128, 16, 16, 256, 3, 2, 1, 1, 0),
    (2, 256, 8, 8, 256, 3, 1, 1, 1, 1),
    (2, 64, 32, 32, 21, 7, 1, 3, 1, 1),
    (2, 256, 16, 16, 128, 3, 2, 1, 1, 0),   # as dgrad: the ConvTranspose forward 8x8 -> 16x16
    (2, 128, 32, 32, 64, 3, 2, 1, 1, 0),
    (2, 256, 10, 10, 256, 3, 1, 0, 1, 0),
    (2, 64, 38, 38, 21, 7, 1, 0, 1, 0),
    (2, 3, 38, 38, 64, 7, 1, 0, 1, 0),
    (2, 21, 38, 38, 64, 7, 1, 0, 1, 0),
    (8, 256, 33, 33, 256, 3, 1, 2, 2, 0),
    (8, 256, 33, 33, 1024, 1, 1, 0, 1, 0),
    (8, 256, 65, 65, 512, 1, 2, 0, 1, 0),
]


def rel(a, b):
    return ((a.double().cpu() - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def main():
    g = torch.Generator().manual_seed(3)
    for (N, C, H, W, K, R, s, p, d, refl) in SHAPES:
        x = torch.randn(N, C, H, W, generator=g, dtype=torch.float64)
        w = torch.randn(K, C, R, R, generator=g, dtype=torch.float64) * 0.05
        xr = x.clone().requires_grad_(True)
        xp = TF.pad(xr, (p, p, p, p), mode="reflect") if refl else xr
        yr = TF.conv2d(xp, w, None, s, 0 if refl else p, d)
        gy = torch.randn(yr.shape, generator=g, dtype=torch.float64)
        yr.backward(gy)
        xg = x.float().to(dev).contiguous(memory_format=CL)
        wg = w.float().to(dev).contiguous(memory_format=CL)
        y = F.conv2d_fwd(xg, wg, None, s, p, d, F.PAD_REFLECT if refl else F.PAD_ZEROS)
        line = "%dx%dx%d c%d k%d r%d s%d p%d d%d refl%d: fwd %.1e" % (N, H, W, C, K, R, s, p, d, refl, rel(y, yr.detach()))
        if not refl:
            wt = F.weight_transposed(wg)
            dx = F.conv2d_dgrad(gy.float().to(dev).contiguous(memory_format=CL), wt, xg.shape, wg.shape, s, p, d)
            line += "  dgrad %.1e" % rel(dx, xr.grad)
        if R == 3 and s == 2:   # ConvTranspose2d(K -> C, 3, stride 2, padding 1, output_padding 1) on the conv's output shape
            wt_ = torch.randn(K, C, R, R, generator=g, dtype=torch.float64) * 0.05
            b_ = torch.randn(C, generator=g, dtype=torch.float64)
            xin = torch.randn(yr.shape, generator=g, dtype=torch.float64)
            ref = torch.relu(TF.conv_transpose2d(xin, wt_, b_, 2, 1, 1))
            out = F.conv_transpose2d(xin.float().to(dev).contiguous(memory_format=CL), wt_.float().to(dev).contiguous(memory_format=CL),
                                     b_.float().to(dev), 2, 1, 1, F.ACT_RELU, 0.0)
            line += "  convT %.1e" % rel(out, ref)
        print(line)


if __name__ == "__main__":
    main()
