#!/usr/bin/env python
"""Per-shape timing of the conv kernels on the MI355X (tuning aid; not part of the product or the tests).
usage: python tools/conv_bench.py [cfgs...]   - times forward/dgrad under each forced tile config and wgrad."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
F = importlib.import_module("semi-supervised-segmentation-cyclegan_amd.functional")
dev = torch.device("cuda:0")
CL = torch.channels_last

# N, C, H, W, K, R, stride, pad, dil
SHAPES = [
    (16, 64, 256, 256, 3, 7, 1, 3, 1),
    (8, 2048, 33, 33, 3, 3, 1, 6, 6),
    (8, 21, 256, 256, 64, 7, 1, 3, 1),
    (8, 64, 256, 256, 21, 7, 1, 3, 1),
    (8, 64, 256, 256, 3, 7, 1, 3, 1),
    (8, 21, 256, 256, 64, 7, 2, 3, 1),
    (8, 256, 33, 33, 256, 3, 1, 2, 2),
    (8, 512, 33, 33, 512, 3, 1, 4, 4),
    (8, 256, 33, 33, 1024, 1, 1, 0, 1),
    (8, 1024, 33, 33, 256, 1, 1, 0, 1),
    (8, 512, 33, 33, 2048, 1, 1, 0, 1),
    (8, 256, 64, 64, 256, 3, 1, 1, 1),
    (8, 64, 65, 65, 64, 3, 1, 1, 1),
    (8, 128, 33, 33, 128, 3, 1, 1, 1),
    (8, 2048, 33, 33, 21, 3, 1, 6, 6),
    (8, 64, 256, 256, 128, 1, 1, 0, 1),
]


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


def main():
    cfgs = [int(a, 0) for a in sys.argv[1:]] or [-1, 0, 1, 2, 3, 4]      # forced tile classes of the exact-fp32 family (-1 = the planner)
    F.set_conv_precision("f32x")
    for (N, C, H, W, K, R, s, p, d) in SHAPES:
        x = torch.randn(N, C, H, W, device=dev).contiguous(memory_format=CL)
        w = (torch.randn(K, C, R, R, device=dev) * 0.05).contiguous(memory_format=CL)
        y = F.conv2d_fwd(x, w, None, s, p, d)
        wt = F.weight_transposed(w)
        gy = torch.randn_like(y)
        flops = 2.0 * y.numel() * C * R * R
        line = "%-44s" % ("%dx%dx%d c%d k%d r%d s%d d%d" % (N, H, W, C, K, R, s, d))
        for cfg in cfgs:
            F.tuning(tile_class=None if cfg < 0 else cfg & 0xff, split=(cfg >> 8) & 0xff if cfg >= 0 else 0)
            try:
                tf = timeit(lambda: F.conv2d_fwd(x, w, None, s, p, d))
                td = timeit(lambda: F.conv2d_dgrad(gy, wt, x.shape, w.shape, s, p, d))
                line += " | %5s f %5.1f d %5.1f" % (hex(cfg) if cfg > 255 else cfg, flops / tf / 1e12, flops / td / 1e12)
            except Exception as e:
                line += " | cfg%2d ERR" % cfg
        F.tuning()
        tw = timeit(lambda: F.conv2d_wgrad(x, gy, w.shape, s, p, d))
        line += " | wg %5.1f" % (flops / tw / 1e12)
        print(line)
        sys.stdout.flush()


if __name__ == "__main__":
    main()
