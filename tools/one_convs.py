#!/usr/bin/env python
"""Run ONE conv shape in the split mode (forward + data gradient) a few times: the subject of a PMC pass.
usage: python tools/one_convs.py N C H W K R stride pad dil [tuning] [iters] [mode]"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
F = importlib.import_module("semi-supervised-segmentation-cyclegan_amd.functional")
N, C, H, W, K, R, s, p, d = [int(v) for v in sys.argv[1:10]]
tun = int(sys.argv[10], 0) if len(sys.argv) > 10 else 0
iters = int(sys.argv[11]) if len(sys.argv) > 11 else 5
mode = sys.argv[12] if len(sys.argv) > 12 else "f32s"
dev = torch.device("cuda:0")
x = torch.randn(N, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
w = (torch.randn(K, C, R, R, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
F.set_conv_precision(mode)
F.TUNING[0] = tun
y = F.conv2d_fwd(x, w, None, s, p, d)
gy = torch.randn_like(y)
for _ in range(iters):
    F.conv2d_fwd(x, w, None, s, p, d)
torch.cuda.synchronize()
