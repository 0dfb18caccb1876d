#!/usr/bin/env python
"""Run ONE conv shape (forward, data gradient, weight gradient) a few times: the subject of a PMC pass.
usage: python tools/one_conv.py N C H W K R stride pad dil [cfg] [iters]"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
F = importlib.import_module("semi-supervised-segmentation-cyclegan_amd.functional")
N, C, H, W, K, R, s, p, d = [int(v) for v in sys.argv[1:10]]
cfg = int(sys.argv[10], 0) if len(sys.argv) > 10 else -1
iters = int(sys.argv[11]) if len(sys.argv) > 11 else 5
dev = torch.device("cuda:0")
x = torch.randn(N, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
w = (torch.randn(K, C, R, R, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
F.lib.sscg_debug_set_conv_cfg(cfg)
y = F.conv2d_fwd(x, w, None, s, p, d)
gy = torch.randn_like(y)
wt = F.weight_transposed(w)
for _ in range(iters):
    F.conv2d_fwd(x, w, None, s, p, d)
    F.conv2d_dgrad(gy, wt, x.shape, w.shape, s, p, d)
    F.conv2d_wgrad(x, gy, w.shape, s, p, d)
torch.cuda.synchronize()
