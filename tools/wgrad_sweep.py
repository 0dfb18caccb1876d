#!/usr/bin/env python
"""Sweep (tile class, pixel splits) of the wgrad kernel per shape on the MI355X (tuning aid for plan_wgrad)."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
F = importlib.import_module("semi-supervised-segmentation-cyclegan_amd.functional")
dev = torch.device("cuda:0")
CL = torch.channels_last

# N, C, H, W, K, R, stride, pad, dil   (the wgrad shapes of the VOC 256x256 step, heaviest first)
SHAPES = [
    (8, 256, 33, 33, 256, 3, 1, 2, 2),
    (8, 512, 33, 33, 512, 3, 1, 4, 4),
    (8, 256, 33, 33, 1024, 1, 1, 0, 1),
    (8, 1024, 33, 33, 256, 1, 1, 0, 1),
    (8, 512, 33, 33, 2048, 1, 1, 0, 1),
    (8, 2048, 33, 33, 512, 1, 1, 0, 1),
    (8, 1024, 33, 33, 2048, 1, 1, 0, 1),
    (8, 128, 33, 33, 128, 3, 1, 1, 1),
    (8, 64, 65, 65, 64, 3, 1, 1, 1),
    (8, 64, 65, 65, 256, 1, 1, 0, 1),
    (8, 128, 33, 33, 512, 1, 1, 0, 1),
    (8, 512, 33, 33, 128, 1, 1, 0, 1),
    (8, 256, 65, 65, 64, 1, 1, 0, 1),
    (8, 256, 64, 64, 256, 3, 1, 1, 1),
    (8, 128, 128, 128, 128, 3, 1, 1, 1),
    (8, 64, 256, 256, 128, 1, 1, 0, 1),
]


def timeit(fn, iters=10):
    for _ in range(2):
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
    only = [a for a in sys.argv[1:]]          # e.g. `r1` = the 1x1 shapes only, `n16` = batch 16
    shapes = [sh for sh in SHAPES if ("r1" not in only or sh[5] == 1)]
    if "n16" in only:
        shapes = [(16,) + sh[1:] for sh in shapes]
    for (N, C, H, W, K, R, s, p, d) in shapes:
        x = torch.randn(N, C, H, W, device=dev).contiguous(memory_format=CL)
        w = (torch.randn(K, C, R, R, device=dev) * 0.05).contiguous(memory_format=CL)
        y = F.conv2d_fwd(x, w, None, s, p, d)
        gy = torch.randn_like(y)
        flops = 2.0 * y.numel() * C * R * R
        steps = (y.numel() // K + 31) // 32
        F.tuning()
        t0 = timeit(lambda: F.conv2d_wgrad(x, gy, w.shape, s, p, d))
        print("%dx%dx%d c%d k%d r%d: steps %d default %.1f TF/s" % (N, H, W, C, K, R, steps, flops / t0 / 1e12))
        for cfg, b in ((0, 128), (1, 64)):
            tiles = ((K + b - 1) // b) * ((R * R * C + b - 1) // b)
            res = []
            cand = sorted(set(max(1, min(steps // 2, int(round(wgs / tiles)))) for wgs in
                              (128, 192, 256, 320, 384, 448, 512, 640, 768, 896, 1024, 1280, 1536, 2048, 3072, 4096)))
            for sp in cand:
                if sp > 256:
                    continue
                F.tuning(wgrad_class=cfg, wgrad_splits=sp)
                try:
                    t = timeit(lambda: F.conv2d_wgrad(x, gy, w.shape, s, p, d))
                except Exception as e:
                    print("   ERR", cfg, sp, e)
                    continue
                res.append((flops / t / 1e12, sp, sp * tiles))
            print("   cfg%d tiles %4d: " % (cfg, tiles) + " ".join("%d/%d:%.0f" % (sp, wg, tf) for tf, sp, wg in res)
                  + "   best %s" % (max(res)[1:],))
        F.tuning()
        t0 = timeit(lambda: F.conv2d_wgrad(x, gy, w.shape, s, p, d))
        print("   cost model again: %.1f TF/s" % (flops / t0 / 1e12))
        sys.stdout.flush()
    F.tuning()


if __name__ == "__main__":
    main()
