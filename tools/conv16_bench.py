#!/usr/bin/env python
"""bf16 conv kernel variants on the shapes of BASELINE config 3 (tuning aid; not part of the product).
usage: python tools/conv16_bench.py [out.txt]"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

F = importlib.import_module("semi-supervised-segmentation-cyclegan_amd.functional")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
F.set_conv_precision("bf16")
CL = torch.channels_last
# (N, C, H, W, K, R, stride, pad, dil)
SHAPES = [(32, 256, 64, 128, 256, 3, 1, 1, 1), (16, 256, 33, 65, 256, 3, 1, 2, 2), (32, 256, 33, 65, 256, 3, 1, 2, 2),
          (16, 512, 33, 65, 512, 3, 1, 4, 4), (16, 256, 33, 65, 1024, 1, 1, 0, 1), (16, 1024, 33, 65, 256, 1, 1, 0, 1),
          (16, 512, 33, 65, 2048, 1, 1, 0, 1), (16, 64, 256, 512, 128, 1, 1, 0, 1), (16, 64, 65, 129, 256, 1, 1, 0, 1),
          (32, 64, 128, 256, 128, 3, 2, 1, 1), (32, 128, 64, 128, 256, 3, 2, 1, 1), (16, 64, 128, 256, 128, 4, 2, 1, 1),
          (16, 128, 64, 128, 256, 4, 2, 1, 1), (16, 256, 32, 64, 512, 4, 1, 1, 1)]
# name, forced tile class (100 + cfg; 0xff = planner), tune flags (1 = LDS-DMA pieces spread over the MFMA groups)
VARIANTS = [("plan", None, 0), ("128x128w8", 4, 0), ("64x64", 1, 0), ("128x128", 0, 0), ("128x64", 3, 0)]      # (name, forced tile class or None, -)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


out = open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout
for (n, c, h, w, k, r, s, p, d) in SHAPES:
    x = torch.randn(n, c, h, w, device=dev).to(torch.bfloat16).contiguous(memory_format=CL)
    wt = (torch.randn(k, c, r, r, device=dev) * 0.05).to(torch.bfloat16).contiguous(memory_format=CL)
    y = F.conv2d_fwd(x, wt, None, s, p, d, out_f32=False)
    gy = torch.randn_like(y.float()).to(torch.bfloat16).contiguous(memory_format=CL)
    wtt = F.weight_transposed(wt, torch.bfloat16)
    flops = 2.0 * n * y.shape[2] * y.shape[3] * k * c * r * r
    row = "%-36s" % ("%dx%dx%d c%d k%d r%d d%d" % (n, h, w, c, k, r, d))
    for _ in range(10):      # clocks up before the first variant is timed
        F.conv2d_fwd(x, wt, None, s, p, d, out_f32=False)
    y_ref = y.float()
    dx_ref = F.conv2d_dgrad(gy, wtt, x.shape, wt.shape, s, p, d, out_dtype=torch.bfloat16).float()
    for name, cfg, tune in VARIANTS:
        old = F.tuning(tile_class=cfg)
        try:
            ey = float((F.conv2d_fwd(x, wt, None, s, p, d, out_f32=False).float() - y_ref).abs().max())
            ed = float((F.conv2d_dgrad(gy, wtt, x.shape, wt.shape, s, p, d, out_dtype=torch.bfloat16).float() - dx_ref).abs().max())
            if ey > 0.0 or ed > 0.0:      # the tile class does not change the order of a row's k-sum: results are bitwise equal (tail splits aside)
                row += " | %s DIFF %.3g %.3g" % (name, ey, ed)
            tf = timeit(lambda: F.conv2d_fwd(x, wt, None, s, p, d, out_f32=False))
            td = timeit(lambda: F.conv2d_dgrad(gy, wtt, x.shape, wt.shape, s, p, d, out_dtype=torch.bfloat16))
            row += " | %-9s f %6.1f d %6.1f" % (name, flops / tf / 1e9, flops / td / 1e9)
        finally:
            F.TUNING[0], F.WGRAD_TUNING[0] = old
    # weight gradient: LDS-DMA + transpose-read kernel, 8 waves (default) / 4 waves, pixel-split variants
    dw_ref = None
    for name, tune in (("wg8-768", 0), ("wg4", 2), ("wg8-512", 4 << 4), ("wg8-1024", 8 << 4), ("wg8-1536", 12 << 4), ("wg8-384", 3 << 4)):
        old = F.tuning(wgrad_flags=tune)
        try:
            dw = F.conv2d_wgrad(x, gy, wt.shape, s, p, d)
            if dw_ref is None:
                dw_ref = dw
            else:
                err = float((dw - dw_ref).abs().max() / dw_ref.abs().max())
                if err > 2e-4:
                    row += " | %s DIFF %.3g" % (name, err)
            tw = timeit(lambda: F.conv2d_wgrad(x, gy, wt.shape, s, p, d))
            row += " | %s %6.1f" % (name, flops / tw / 1e9)
        finally:
            F.TUNING[0], F.WGRAD_TUNING[0] = old
    out.write(row + "\n")
    out.flush()
