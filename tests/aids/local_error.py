#!/usr/bin/env python
"""Where is the build's DeepLab forward noisier than the reference's fp32 arithmetic?  (forward rel-L2 against
fp64 3.2e-4 / 6.3e-4 for the build's two fp32 arithmetics vs 2.7e-4 / 4.4e-4 for torch's CPU kernels - a stable ratio, not a draw:
the reference's own figure moves by < 10 % over thread counts and 1-ulp weight perturbations.)

Per-OPERATION local errors: every conv / normalisation call of one HIP forward is re-computed from ITS OWN fp32 inputs (teacher
forcing at operation level) in fp64 (truth) and by torch's CPU fp32 kernel (the reference's arithmetic), so the two arithmetics are
compared on identical inputs, one rounding-error source at a time.  GPU box only; debug aid (not a test).

usage: python tests/aids/local_error.py [deeplab_3_21|deeplab_21_3] [f32s|f32x]"""
import importlib
import os
import sys

import torch
import torch.nn.functional as TF

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
PKG = "semi-supervised-segmentation-cyclegan_amd"
F = importlib.import_module(PKG + ".functional")
arch = importlib.import_module(PKG + ".arch")
from oracle import fixtures as FX  # noqa: E402  (debug tool only)

dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "deeplab_3_21"
mode = sys.argv[2] if len(sys.argv) > 2 else "f32s"
_, kind, args, xshape = [n for n in FX.NETS if n[0] == name][0]
F.set_conv_precision(mode)
rows = []
state = {}


def rl2(a, b):
    return float((a.double() - b).norm() / b.norm().clamp_min(1e-300))


orig = {k: getattr(F, k) for k in ("conv2d_fwd", "norm_stats_from_conv", "norm_stats", "norm_apply")}


def conv2d_fwd(x, w, bias, stride=1, pad=0, dil=1, pad_mode=F.PAD_ZEROS, act=F.ACT_NONE, slope=0.0, out_f32=True, stats=None):
    out = orig["conv2d_fwd"](x, w, bias, stride, pad, dil, pad_mode, act, slope, out_f32, stats)
    y = out[0] if isinstance(out, tuple) else out
    xc, wc = F.to_nchw(x).cpu().contiguous(), F.to_nchw(w).cpu().contiguous()
    bc = None if bias is None else bias.detach().cpu()
    if pad_mode != F.PAD_ZEROS:
        xc, p = TF.pad(xc, (pad,) * 4, mode="reflect"), 0
    else:
        p = pad
    r64 = TF.conv2d(xc.double(), wc.double(), None if bc is None else bc.double(), stride, p, dil)
    r32 = TF.conv2d(xc, wc, bc, stride, p, dil)
    yh = F.to_nchw(y).cpu()
    tag = "conv%dx%d c%d k%d%s" % (w.shape[2], w.shape[3], w.shape[1], w.shape[0], " d%d" % dil if dil > 1 else "")
    rows.append(("conv%dx%d" % (w.shape[2], w.shape[3]) if w.shape[1] >= 32 and w.shape[0] >= 32 else "conv-thin", tag, rl2(yh, r64), rl2(r32, r64)))
    state["y"] = y
    return out


def _after_stats(mean, rstd, eps):
    y = F.to_nchw(state["y"]).cpu().double()          # BatchNorm over (N, H, W) of the conv output just produced
    m64 = y.mean((0, 2, 3))
    v64 = y.var((0, 2, 3), unbiased=False)
    r64 = 1.0 / torch.sqrt(v64 + eps)
    if mean.shape[0] == 1:
        em = float(((mean[0].cpu().double() - m64) * r64).abs().max())      # error of the mean in units of the channel's std
        er = float(((rstd[0].cpu().double() - r64) / r64).abs().max())
        rows.append(("bn-stats", "c%d" % mean.shape[1], em, er))


def norm_stats_from_conv(cs, glc, eps, running_mean=None, running_var=None, momentum=0.1):
    mean, rstd = orig["norm_stats_from_conv"](cs, glc, eps, running_mean, running_var, momentum)
    _after_stats(mean, rstd, eps)
    return mean, rstd


def norm_stats(x, per_sample, eps=1e-5, running_mean=None, running_var=None, momentum=0.1):
    mean, rstd = orig["norm_stats"](x, per_sample, eps, running_mean, running_var, momentum)
    state["y"] = x
    _after_stats(mean, rstd, eps)
    return mean, rstd


def norm_apply(x, mean, rstd, gamma, beta, residual, per_sample, act=F.ACT_NONE, slope=0.0):
    z = orig["norm_apply"](x, mean, rstd, gamma, beta, residual, per_sample, act, slope)
    if per_sample is False:
        xc = F.to_nchw(x).cpu().contiguous()
        g, b = (None, None) if gamma is None else (gamma.detach().cpu(), beta.detach().cpu())
        rc = None if residual is None else F.to_nchw(residual).cpu().contiguous()

        def unit(dt):
            o = TF.batch_norm(xc.to(dt), None, None, None if g is None else g.to(dt), None if b is None else b.to(dt), True, 0.1, 1e-5)
            if rc is not None:
                o = o + rc.to(dt)
            return torch.relu(o) if act == F.ACT_RELU else o
        z64, z32 = unit(torch.float64), unit(torch.float32)
        rows.append(("bn+res" if rc is not None else "bn", "c%d" % x.shape[1], rl2(F.to_nchw(z).cpu(), z64), rl2(z32, z64)))
    return z


for k, f in (("conv2d_fwd", conv2d_fwd), ("norm_stats_from_conv", norm_stats_from_conv), ("norm_stats", norm_stats), ("norm_apply", norm_apply)):
    setattr(F, k, f)

m = arch.define_Gen(args[0], args[1], 64, kind, norm="instance", use_dropout=False, gpu_ids=[0])
m.load_state_dict(FX.net_weights(name, kind, args), strict=True)
m.train()
with torch.no_grad():
    y = m(FX.net_input(name, xshape).to(dev))
torch.cuda.synchronize()
verbose = os.environ.get("VERBOSE", "0") == "1"
by = {}
for cls, tag, eh, et in rows:
    by.setdefault(cls, []).append((eh, et))
    if verbose:
        print("%-10s %-26s hip %.2e  torch-cpu-fp32 %.2e  ratio %.2f" % (cls, tag, eh, et, eh / max(et, 1e-300)))
print("%s, mode %s: local error of each operation on its own inputs, rms over calls (hip | torch CPU fp32 | ratio)" % (name, mode))
for cls, v in by.items():
    t = torch.tensor(v, dtype=torch.float64)
    rh, rt = float(t[:, 0].square().mean().sqrt()), float(t[:, 1].square().mean().sqrt())
    if cls == "bn-stats":
        print("  %-10s x%-3d mean error / std: rms %.2e max %.2e   rstd rel error: rms %.2e max %.2e" % (cls, len(v), rh, float(t[:, 0].max()), rt, float(t[:, 1].max())))
    else:
        print("  %-10s x%-3d %.2e | %.2e | %.2f" % (cls, len(v), rh, rt, rh / rt))
