"""Kernel-level parity: every HIP kernel (through the C ABI) against the torch CPU op the reference
calls at that site, evaluated in fp64.  Tolerances are relative to the largest reference magnitude."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as TF

from conftest import load_sub

pytestmark = pytest.mark.gpu

CL = torch.channels_last


def rel_err(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def gpu(t, dev):
    return t.float().to(dev).contiguous(memory_format=CL) if t.dim() == 4 else t.float().to(dev)


# (N, C, H, W, K, R, stride, pad, dil, reflect, bias, act)
CONV_CASES = [
    (2, 256, 33, 33, 256, 3, 1, 2, 2, 0, 0, 0),    # DeepLab layer3 conv2 (generators.py:331, dilation 2)
    (2, 512, 17, 17, 512, 3, 1, 4, 4, 0, 0, 0),    # layer4 conv2
    (2, 64, 33, 33, 64, 3, 1, 1, 1, 0, 0, 0),      # layer1 conv2
    (2, 3, 64, 64, 64, 7, 2, 3, 1, 0, 0, 0),       # conv1 (3-channel image)
    (2, 21, 64, 64, 64, 7, 2, 3, 1, 0, 0, 0),      # conv1 (21-channel one-hot)
    (2, 2048, 9, 9, 21, 3, 1, 6, 6, 0, 1, 0),      # classifier, dilation 6, bias
    (2, 2048, 9, 9, 3, 3, 1, 12, 12, 0, 1, 0),     # classifier, dilation 12 (pad > feature map)
    (2, 256, 33, 33, 512, 1, 2, 0, 1, 0, 0, 0),    # 1x1 stride-2 downsample
    (2, 1024, 17, 17, 256, 1, 1, 0, 1, 0, 0, 0),   # 1x1 bottleneck
    (2, 64, 32, 32, 128, 3, 2, 1, 1, 0, 1, 0),     # ResnetGenerator down conv
    (2, 256, 16, 16, 256, 3, 1, 1, 1, 1, 1, 0),    # ResidualBlock conv, reflection pad 1
    (2, 21, 32, 32, 64, 7, 1, 3, 1, 1, 1, 0),      # ResnetGenerator stem, reflection pad 3
    (2, 64, 32, 32, 3, 7, 1, 3, 1, 1, 1, 3),       # ResnetGenerator head + tanh
    (2, 3, 32, 32, 64, 1, 1, 0, 1, 0, 1, 2),       # PixelDiscriminator conv1 + LeakyReLU
    (2, 128, 32, 32, 1, 1, 1, 0, 1, 0, 1, 0),      # PixelDiscriminator conv3
    (2, 21, 32, 32, 64, 4, 2, 1, 1, 0, 1, 2),      # PatchGAN conv1
    (2, 128, 16, 16, 256, 4, 2, 1, 1, 0, 1, 0),    # PatchGAN conv3
    (2, 256, 9, 9, 512, 4, 1, 1, 1, 0, 1, 0),      # PatchGAN conv4 (stride 1)
    (1, 20, 17, 19, 36, 3, 1, 1, 1, 0, 1, 1),      # ragged everything + relu
    # bench-size maps: whole rounds of tiles unsplit + a K-split tail (plan_kc_split), cost-model wgrad plans
    (8, 256, 33, 33, 256, 3, 1, 2, 2, 0, 1, 0),    # 548 tiles = 512 whole + 36 split; bias in the tail reduce (no ReLU: 2.2 M
                                                   # outputs always hold a few within fp32 rounding of 0, whose mask flips vs fp64)
    (8, 256, 33, 33, 1024, 1, 1, 0, 1, 0, 0, 0),   # 1x1 expansion, 137 x 16 tiles
    (8, 64, 65, 65, 64, 3, 1, 1, 1, 0, 0, 0),      # 529 tiles = 512 + 17
]


def ref_conv(x, w, b, stride, pad, dil, reflect, act):
    if reflect:
        x = TF.pad(x, (pad, pad, pad, pad), mode="reflect")
        pad = 0
    y = TF.conv2d(x, w, b, stride, pad, dil)
    if act == 1:
        y = torch.relu(y)
    elif act == 2:
        y = TF.leaky_relu(y, 0.2)
    elif act == 3:
        y = torch.tanh(y)
    return y


@pytest.mark.parametrize("mode", ["f32x", "f32s"])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_bwd(case, mode, F, dev):
    """nn.Conv2d forward / input gradient / weight gradient / bias gradient through the autograd function, against torch fp64, in
    both contractions of fp32 tensors: the exact fp32 MFMA family ("f32x") and the split contraction ("f32s" = --dtype f32)."""
    F.set_conv_precision(mode)
    try:
        _conv_fwd_bwd(case, F, dev)
    finally:
        F.set_conv_precision("f32")


def _conv_fwd_bwd(case, F, dev):
    N, C, H, W, K, R, stride, pad, dil, reflect, bias, act = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn(N, C, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(K, C, R, R, generator=g, dtype=torch.float64) * 0.05
    b = torch.randn(K, generator=g, dtype=torch.float64) if bias else None
    xr = x.clone().requires_grad_(not reflect)
    wr = w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if bias else None
    yr = ref_conv(xr, wr, br, stride, pad, dil, reflect, act)
    gy = torch.randn(yr.shape, generator=g, dtype=torch.float64)
    yr.backward(gy)

    xg = gpu(x, dev).requires_grad_(not reflect)
    wg = gpu(w, dev).requires_grad_(True)
    bg = gpu(b, dev).requires_grad_(True) if bias else None
    yg = F.conv2d(xg, wg, bg, stride, pad, dil, 1 if reflect else 0, act, 0.2)
    assert tuple(yg.shape) == tuple(yr.shape)
    assert rel_err(yg, yr) < 2e-5
    yg.backward(gpu(gy, dev))
    assert rel_err(wg.grad, wr.grad) < 5e-5
    if bias:
        assert rel_err(bg.grad, br.grad) < 2e-5
    if not reflect:
        assert rel_err(xg.grad, xr.grad) < 2e-5


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 5, 6, 7])
def test_conv_all_tile_configs(cfg, F, dev):
    """Every tile class of the exact-fp32 kernel family, forced through sscg_conv_desc.tuning."""
    g = torch.Generator().manual_seed(cfg)
    x = torch.randn(2, 64, 21, 23, generator=g, dtype=torch.float64)
    w = torch.randn(96, 64, 3, 3, generator=g, dtype=torch.float64) * 0.05
    yr = TF.conv2d(x, w, None, 1, 1, 1)
    F.set_conv_precision("f32x")
    old = F.tuning(tile_class=cfg)
    try:
        yg = F.conv2d_fwd(gpu(x, dev), gpu(w, dev), None, 1, 1, 1)
        gy = torch.randn(yr.shape, generator=g, dtype=torch.float64)
        wt = F.weight_transposed(gpu(w, dev))
        dx = F.conv2d_dgrad(gpu(gy, dev), wt, x.shape, w.shape, 1, 1, 1)
    finally:
        F.TUNING[0], F.WGRAD_TUNING[0] = old
        F.set_conv_precision("f32")
    assert rel_err(yg, yr) < 2e-5
    dxr = torch.autograd.grad(TF.conv2d(x.requires_grad_(True), w, None, 1, 1, 1), x, gy)[0]
    assert rel_err(dx, dxr) < 2e-5


def test_conv_mfma_layout_asymmetric(F, dev):
    """A = identity-like probe with an asymmetric weight catches transposed fragment layouts."""
    C = K = 64
    x = torch.zeros(1, C, 8, 8, dtype=torch.float64)
    for c in range(C):
        x[0, c, c % 8, (c // 8) % 8] = 1.0 + c
    w = torch.arange(K * C, dtype=torch.float64).reshape(K, C, 1, 1) / (K * C)
    yr = TF.conv2d(x, w)
    yg = F.conv2d_fwd(gpu(x, dev), gpu(w, dev), None)
    assert rel_err(yg, yr) < 1e-6


@pytest.mark.parametrize("geom", [(2, 256, 16, 16, 128, 1), (2, 128, 16, 16, 64, 1), (1, 64, 9, 11, 24, 0)])
def test_conv_transpose(geom, F, dev):
    N, Cin, H, W, Cout, bias = geom
    g = torch.Generator().manual_seed(7)
    x = torch.randn(N, Cin, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(Cin, Cout, 3, 3, generator=g, dtype=torch.float64) * 0.05
    b = torch.randn(Cout, generator=g, dtype=torch.float64) if bias else None
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if bias else None
    yr = TF.conv_transpose2d(xr, wr, br, 2, 1, 1)
    gy = torch.randn(yr.shape, generator=g, dtype=torch.float64)
    yr.backward(gy)
    xg, wg = gpu(x, dev).requires_grad_(True), gpu(w, dev).requires_grad_(True)
    bg = gpu(b, dev).requires_grad_(True) if bias else None
    yg = F.conv_transpose2d(xg, wg, bg, 2, 1, 1)
    assert tuple(yg.shape) == tuple(yr.shape)
    assert rel_err(yg, yr) < 2e-5
    yg.backward(gpu(gy, dev))
    assert rel_err(xg.grad, xr.grad) < 2e-5
    assert rel_err(wg.grad, wr.grad) < 5e-5
    if bias:
        assert rel_err(bg.grad, br.grad) < 2e-5


@pytest.mark.parametrize("per_sample", [True, False])
@pytest.mark.parametrize("act", [0, 1, 2])
@pytest.mark.parametrize("shape", [(2, 64, 17, 19), (3, 256, 9, 9), (2, 2048, 5, 5), (2, 20, 8, 8)])
def test_norm_act(per_sample, act, shape, F, dev):
    N, C, H, W = shape
    g = torch.Generator().manual_seed(11)
    x = torch.randn(shape, generator=g, dtype=torch.float64) * 2 + 0.7
    res = torch.randn(shape, generator=g, dtype=torch.float64)
    gamma = torch.randn(C, generator=g, dtype=torch.float64) * 0.1 + 1
    beta = torch.randn(C, generator=g, dtype=torch.float64) * 0.1
    rm0 = torch.randn(C, generator=g, dtype=torch.float64) * 0.1
    rv0 = torch.rand(C, generator=g, dtype=torch.float64) + 0.5
    xr, rr = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
    gr, btr = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm, rv = rm0.clone(), rv0.clone()
    if per_sample:
        yr = TF.instance_norm(xr, eps=1e-5) + rr
    else:
        yr = TF.batch_norm(xr, rm, rv, gr, btr, True, 0.1, 1e-5) + rr
    if act == 1:
        yr = torch.relu(yr)
    elif act == 2:
        yr = TF.leaky_relu(yr, 0.2)
    gy = torch.randn(shape, generator=g, dtype=torch.float64)
    yr.backward(gy)

    xg, rg = gpu(x, dev).requires_grad_(True), gpu(res, dev).requires_grad_(True)
    if per_sample:
        yg = F.instance_norm_act(xg, act, 0.2, residual=rg)
    else:
        gg, bg = gpu(gamma, dev).requires_grad_(True), gpu(beta, dev).requires_grad_(True)
        rmg, rvg = gpu(rm0, dev), gpu(rv0, dev)
        yg = F.batch_norm_act(xg, gg, bg, rmg, rvg, True, 0.1, 1e-5, act, 0.2, residual=rg)
    assert rel_err(yg, yr) < 1e-5
    yg.backward(gpu(gy, dev))
    assert rel_err(xg.grad, xr.grad) < 2e-5
    assert rel_err(rg.grad, rr.grad) < 1e-6
    if not per_sample:
        assert rel_err(gg.grad, gr.grad) < 1e-5
        assert rel_err(bg.grad, btr.grad) < 1e-5
        assert rel_err(rmg, rm) < 1e-6
        assert rel_err(rvg, rv) < 1e-6


def test_bn_eval_and_frozen_affine(F, dev):
    shape = (2, 64, 9, 9)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(shape, generator=g, dtype=torch.float64)
    gamma = torch.randn(64, generator=g, dtype=torch.float64) * 0.1 + 1
    beta = torch.randn(64, generator=g, dtype=torch.float64) * 0.1
    rm = (torch.randn(64, generator=g) * 0.1).double()   # fp32-representable: compared bit-exactly below
    rv = (torch.rand(64, generator=g) + 0.5).double()
    xr = x.clone().requires_grad_(True)
    yr = torch.relu(TF.batch_norm(xr, rm.clone(), rv.clone(), gamma, beta, False, 0.1, 1e-5))
    gy = torch.randn(shape, generator=g, dtype=torch.float64)
    yr.backward(gy)
    xg = gpu(x, dev).requires_grad_(True)
    rmg, rvg = gpu(rm, dev), gpu(rv, dev)
    yg = F.batch_norm_act(xg, gpu(gamma, dev), gpu(beta, dev), rmg, rvg, False, 0.1, 1e-5, 1, 0.0)
    assert rel_err(yg, yr) < 1e-5
    yg.backward(gpu(gy, dev))
    assert rel_err(xg.grad, xr.grad) < 1e-5
    assert rel_err(rmg, rm) == 0.0  # eval mode must not touch running stats


@pytest.mark.parametrize("hw", [(128, 128), (64, 64), (33, 65), (9, 10)])
def test_maxpool(hw, F, dev):
    g = torch.Generator().manual_seed(3)
    x = torch.relu(torch.randn(2, 64, *hw, generator=g)).double()  # fp32-representable, many ties at 0
    xr = x.clone().requires_grad_(True)
    yr = TF.max_pool2d(xr, 3, 2, 1, ceil_mode=True)
    gy = torch.randn(yr.shape, generator=g, dtype=torch.float64)
    yr.backward(gy)
    xg = gpu(x, dev).requires_grad_(True)
    yg = F.MaxPoolFn.apply(xg)
    assert tuple(yg.shape) == tuple(yr.shape)
    assert rel_err(yg, yr) == 0.0
    yg.backward(gpu(gy, dev))
    assert rel_err(xg.grad, xr.grad) < 1e-6


@pytest.mark.parametrize("geom", [(2, 21, 33, 33, 256, 256), (2, 3, 9, 9, 64, 64), (1, 4, 17, 17, 128, 128),
                                  (2, 20, 5, 9, 32, 64), (1, 3, 1, 1, 8, 8), (2, 3, 16, 24, 16, 24)])
def test_upsample(geom, F, dev):
    N, C, H, W, OH, OW = geom
    g = torch.Generator().manual_seed(4)
    x = torch.randn(N, C, H, W, generator=g, dtype=torch.float64)
    xr = x.clone().requires_grad_(True)
    yr = TF.interpolate(xr, size=(OH, OW), mode="bilinear", align_corners=True)
    gy = torch.randn(yr.shape, generator=g, dtype=torch.float64)
    yr.backward(gy)
    xg = gpu(x, dev).requires_grad_(True)
    yg = F.upsample_bilinear(xg, (OH, OW))
    assert rel_err(yg, yr) < 1e-5   # fp32 interpolation weights (torch computes them in fp32 too)
    yg.backward(gpu(gy, dev))
    assert rel_err(xg.grad, xr.grad) < 1e-5


@pytest.mark.parametrize("C", [21, 20, 4])
def test_softmax_ce_losses(C, F, dev):
    g = torch.Generator().manual_seed(6)
    x = torch.randn(2, C, 32, 32, generator=g, dtype=torch.float64) * 3
    lab = torch.randint(0, C, (2, 32, 32), generator=g)
    xr = x.clone().requires_grad_(True)
    sr = torch.softmax(xr, 1)
    gy = torch.randn(x.shape, generator=g, dtype=torch.float64)
    sr.backward(gy)
    xg = gpu(x, dev).requires_grad_(True)
    sg = F.softmax2d(xg)
    assert rel_err(sg, sr) < 1e-6
    sg.backward(gpu(gy, dev))
    assert rel_err(xg.grad, xr.grad) < 1e-5

    xr2 = x.clone().requires_grad_(True)
    lr_ = TF.cross_entropy(xr2, lab)
    (lr_ * 0.37).backward()
    xg2 = gpu(x, dev).requires_grad_(True)
    lg = F.cross_entropy(xg2, lab.to(dev))
    assert rel_err(lg, lr_) < 1e-6
    F.weighted_sum([lg], [0.37]).backward()
    assert rel_err(xg2.grad, xr2.grad) < 1e-5


@pytest.mark.parametrize("branches", ["ce", "soft", "both"])
@pytest.mark.parametrize("geom", [(2, 21, 33, 33, 256, 256), (2, 4, 9, 9, 64, 64), (1, 20, 17, 33, 128, 256), (2, 21, 1, 1, 8, 8),
                                  (1, 7, 5, 3, 40, 37)])
def test_upsample_head_fused(geom, branches, F, dev):
    """interp -> {softmax2d, CrossEntropyLoss} from the low-resolution logits (model.py:390-392, 398, 401-402, 455;
    sscg_upsample_head_fwd / _bwd: the resized logits are never written) against torch in fp64, and against the three separate HIP
    passes; out-of-range labels are ignored as nn.CrossEntropyLoss ignores ignore_index."""
    N, C, H, W, OH, OW = geom
    g = torch.Generator().manual_seed(9)
    x = torch.randn(N, C, H, W, generator=g, dtype=torch.float64) * 2
    lab = torch.randint(0, C, (N, OH, OW), generator=g)
    lab[0, :3] = 255
    lab[-1, OH // 2, :] = -100
    ref_lab = lab.clone()
    ref_lab[(lab < 0) | (lab >= C)] = -100
    gy = torch.randn(N, C, OH, OW, generator=g, dtype=torch.float64)
    want_ce, want_soft = branches in ("ce", "both"), branches in ("soft", "both")

    xr = x.clone().requires_grad_(True)
    up = TF.interpolate(xr, size=(OH, OW), mode="bilinear", align_corners=True)
    lr_ = TF.cross_entropy(up, ref_lab, ignore_index=-100) if want_ce else None
    sr = torch.softmax(up, 1) if want_soft else None
    ((lr_ * 0.37 if want_ce else 0) + ((sr * gy).sum() if want_soft else 0)).backward()

    def run(fused):
        was = F.FUSE_HEAD[0]
        F.FUSE_HEAD[0] = fused
        try:
            xg = gpu(x, dev).requires_grad_(True)
            assert F._head_applies(xg, OH, OW) == fused
            sg, lg = F.upsample_softmax_ce(xg, (OH, OW), lab.to(dev) if want_ce else None, want_soft=want_soft)
            assert (sg is not None) == want_soft and (lg is not None) == want_ce
            terms, wts = [], []
            if want_ce:
                terms.append(lg); wts.append(0.37)
            if want_soft:
                terms.append((sg * gpu(gy, dev)).sum()); wts.append(1.0)
            F.weighted_sum(terms, wts).backward() if len(terms) > 1 or want_ce else terms[0].backward()
            return sg, lg, xg.grad
        finally:
            F.FUSE_HEAD[0] = was

    sg, lg, dx = run(True)
    if want_ce:
        assert rel_err(lg, lr_) < 1e-6
    if want_soft:
        assert rel_err(sg, sr) < 2e-5       # fp32 interpolation weights (test_upsample: 1e-5 on the resized logits themselves)
    assert rel_err(dx, xr.grad) < 3e-5
    sg0, lg0, dx0 = run(False)          # the separate passes (upsample, softmax, cross entropy) agree to fp32 rounding
    if want_ce:
        assert rel_err(lg, lg0) < 1e-6
    if want_soft:
        assert rel_err(sg, sg0) < 1e-6
    assert rel_err(dx, dx0) < 1e-5


def test_cross_entropy_ignores_out_of_range_labels(F, dev):
    """Labels outside [0, C) (255 'void', -100) follow nn.CrossEntropyLoss's ignore_index semantics: excluded from the mean,
    zero gradient, no out-of-bounds read.  label_onehot writes an all-zero row for them."""
    g = torch.Generator().manual_seed(21)
    C = 21
    x = torch.randn(2, C, 16, 16, generator=g, dtype=torch.float64)
    lab = torch.randint(0, C, (2, 16, 16), generator=g)
    lab[0, :4] = 255
    lab[1, 5, :] = -100
    ref_lab = lab.clone()
    ref_lab[(lab < 0) | (lab >= C)] = -100
    xr = x.clone().requires_grad_(True)
    lr_ = TF.cross_entropy(xr, ref_lab, ignore_index=-100)
    lr_.backward()
    xg = gpu(x, dev).requires_grad_(True)
    lg = F.cross_entropy(xg, lab.to(dev))
    assert rel_err(lg, lr_) < 1e-6
    lg.backward()
    assert rel_err(xg.grad, xr.grad) < 1e-5
    assert float(xg.grad[0, :, :4].abs().max()) == 0.0
    oh = F.label_onehot(lab.unsqueeze(1).to(dev), C).cpu()
    assert float(oh[0, :, :4].abs().max()) == 0.0 and float(oh.sum()) == float(((lab >= 0) & (lab < C)).sum())
    # not one valid pixel: NaN, like torch (a broken label pipeline must not train on a silently empty loss), zero gradient
    xe = gpu(x, dev).requires_grad_(True)
    le = F.cross_entropy(xe, torch.full_like(lab, 255).to(dev))
    assert bool(torch.isnan(le))
    assert bool(torch.isnan(TF.cross_entropy(x, torch.full_like(lab, -100), ignore_index=-100)))


def test_gauss_noise_statistics(F, dev):
    """utils.GaussianNoise (utils.py:116-140): y = x + sigma * x * n with n ~ N(0, 1); torch's generator cannot be matched, so
    the relative noise (y/x - 1)/sigma is checked for zero mean, unit variance, determinism per seed."""
    x = torch.full((4, 3, 128, 128), 2.0, device=dev)
    y1, y2, y3 = F.gauss_noise(x, 0.2, 77), F.gauss_noise(x, 0.2, 77), F.gauss_noise(x, 0.2, 78)
    assert torch.equal(y1, y2) and not torch.equal(y1, y3)
    z = ((y1 / x - 1.0) / 0.2).double().cpu()
    assert abs(float(z.mean())) < 0.02 and abs(float(z.std()) - 1.0) < 0.02
    assert abs(float((z ** 3).mean())) < 0.05 and abs(float((z ** 4).mean()) - 3.0) < 0.15
    utils = __import__("conftest").load_sub("utils")
    gn = utils.GaussianNoise(sigma=0.2)
    a, b = gn(x), gn(x)
    assert not torch.equal(a, b)            # a fresh draw per call
    gn.training = False
    assert gn(x) is x


def test_mse_l1_weighted(F, dev):
    g = torch.Generator().manual_seed(8)
    a = torch.randn(2, 3, 32, 32, generator=g, dtype=torch.float64)
    b = torch.randn(2, 3, 32, 32, generator=g, dtype=torch.float64)
    d = torch.randn(2, 1, 32, 32, generator=g, dtype=torch.float64)
    ar, dr = a.clone().requires_grad_(True), d.clone().requires_grad_(True)
    l1r = (ar - b).abs().mean()
    m1r = ((dr - 1.0) ** 2).mean()
    m0r = (dr ** 2).mean()
    tot_r = 1.0 * l1r + 0.5 * m1r + 2.0 * m0r
    tot_r.backward()
    ag, dg = gpu(a, dev).requires_grad_(True), gpu(d, dev).requires_grad_(True)
    l1g = F.l1_loss(ag, gpu(b, dev))
    m1g = F.mse_const(dg, 1.0)
    m0g = F.mse_const(dg, 0.0)
    tot = F.weighted_sum([l1g, m1g, m0g], [1.0, 0.5, 2.0])
    for u, v in ((l1g, l1r), (m1g, m1r), (m0g, m0r), (tot, tot_r)):
        assert rel_err(u, v) < 1e-6
    tot.backward()
    assert rel_err(ag.grad, ar.grad) < 1e-6
    assert rel_err(dg.grad, dr.grad) < 1e-6


def test_argmax_onehot_bit_exact(F, dev):
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 21, 32, 32, generator=g)
    x[:, 5] = x[:, 3]  # forced ties: the lower index must win (torch.max(dim) rule)
    x[0, :, 0, 0] = 1.0
    idx_r = x.max(1)[1]
    oh_r = torch.zeros_like(x).scatter_(1, idx_r.unsqueeze(1), 1.0)
    oh_g, idx_g = F.argmax_onehot(gpu(x, dev), want_index=True)
    assert torch.equal(idx_g.cpu(), idx_r)
    assert torch.equal(oh_g.cpu().contiguous(), oh_r)
    lab = torch.randint(0, 21, (2, 1, 32, 32), generator=g)
    oh2 = F.label_onehot(lab.to(dev), 21)
    assert torch.equal(oh2.cpu().contiguous(), torch.zeros(2, 21, 32, 32).scatter_(1, lab, 1.0))


def test_adam_matches_torch(F, dev):
    g = torch.Generator().manual_seed(10)
    p0 = torch.randn(10007, generator=g)
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=2e-4, betas=(0.5, 0.999))
    pg = p0.clone().to(dev)
    m = torch.zeros_like(pg)
    v = torch.zeros_like(pg)
    for step in range(1, 4):
        gr = torch.randn(10007, generator=g)
        pr.grad = gr.clone()
        opt.step()
        F.adam_step(pg, gr.to(dev), m, v, 2e-4, 0.5, 0.999, 1e-8, step)
    assert rel_err(pg, pr) < 1e-6
    assert rel_err(m, opt.state[pr]["exp_avg"]) < 1e-6
    assert rel_err(v, opt.state[pr]["exp_avg_sq"]) < 1e-6


def test_layout_roundtrip_and_dropout(F, dev):
    g = torch.Generator().manual_seed(12)
    x = torch.randn(2, 21, 17, 19, generator=g).to(dev)
    y = F.to_nhwc(x)
    assert y.is_contiguous(memory_format=CL) and torch.equal(y.cpu(), x.cpu())
    z = F.to_nchw(y)
    assert z.is_contiguous() and torch.equal(z.cpu(), x.cpu())
    big = torch.ones(4, 64, 32, 32, device=dev).contiguous(memory_format=CL)
    d1 = F.dropout(big, 0.5, 1234)
    d2 = F.dropout(big, 0.5, 1234)
    assert torch.equal(d1, d2)
    keep = (d1 != 0).float().mean().item()
    assert abs(keep - 0.5) < 0.01
    assert set(d1.unique().cpu().tolist()) == {0.0, 2.0}
    rp = F.reflect_pad(y, 3)
    assert torch.equal(rp.cpu().contiguous(), TF.pad(x.cpu(), (3, 3, 3, 3), mode="reflect"))


@pytest.mark.parametrize("C", [21, 20, 4, 1])
def test_confusion_hist_bit_exact(C, F, dev):
    """runningScore._fast_hist (utils.py:363-369): integer counts, void (255) and negative labels ignored."""
    g = torch.Generator().manual_seed(C)
    n = 3 * 65 * 67
    lt = torch.randint(0, C, (n,), generator=g, dtype=torch.int64)
    lt[torch.rand(n, generator=g) < 0.05] = 255
    lt[torch.rand(n, generator=g) < 0.01] = -1
    lp = torch.randint(0, C, (n,), generator=g, dtype=torch.int64)
    keep = (lt >= 0) & (lt < C)
    ref = np.bincount((C * lt[keep] + lp[keep]).numpy(), minlength=C * C).reshape(C, C)
    h = F.confusion_hist(lt.to(dev), lp.to(dev), C)
    h = F.confusion_hist(lt.to(dev), lp.to(dev), C, h)          # accumulates
    assert np.array_equal(h.cpu().numpy(), 2 * ref)


def test_running_score_device_matches_host(F, dev):
    utils = load_sub("utils")
    g = torch.Generator().manual_seed(3)
    host, devs = utils.runningScore(21, "voc2012"), utils.runningScore(21, "voc2012")
    for _ in range(3):
        lt = torch.randint(0, 21, (2, 33, 35), generator=g, dtype=torch.int64)
        lt[0, :4] = 255
        lp = torch.randint(0, 21, (2, 33, 35), generator=g, dtype=torch.int64)
        host.update(lt.numpy(), lp.numpy())
        devs.update_device(lt.to(dev), lp.to(dev))
    (s0, c0), (s1, c1) = host.get_scores(), devs.get_scores()
    assert s0 == s1 and c0 == c1


def _bench_shapes():
    import os
    import re
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bench_conv_shapes.txt")
    out = []
    for line in open(path):
        m = re.match(r"(\d+)x(\d+)x(\d+) c(\d+) k(\d+) r(\d+) s(\d+) p(\d+) d(\d+)", line.strip())
        if m:
            out.append(tuple(int(v) for v in m.groups()))
    return sorted(set(out))


@pytest.mark.parametrize("mode", ["f32x", "f32s"])
@pytest.mark.parametrize("shape", _bench_shapes(), ids=lambda s: "%dx%dx%d_c%d_k%d_r%d_s%d_p%d_d%d" % s)
def test_conv_adjoint_identities_at_bench_size(shape, mode, F, dev):
    """Every convolution shape of the BASELINE step (VOC 256x256, batch 8; list recorded by bench.py) at FULL size:
    <conv(x, w), dy> = <x, dgrad(dy, w)> = <w, wgrad(x, dy)>.  Size-independent, ties the three kernels (and their split /
    parity-class / tile plans at these sizes) to each other; the inner products are taken in fp64.  Both fp32 kernel families:
    the exact fp32 MFMA ("f32x") and the split contraction ("f32s", what --dtype f32 runs)."""
    F.set_conv_precision(mode)
    try:
        _adjoint_at_bench_size(shape, F, dev)
    finally:
        F.set_conv_precision("f32")


def _adjoint_at_bench_size(shape, F, dev):
    N, H, W, C, K, R, s, p, d = shape
    g = torch.Generator(device=dev).manual_seed(sum(shape))
    x = torch.randn(N, C, H, W, device=dev, generator=g).contiguous(memory_format=CL)
    w = (torch.randn(K, C, R, R, device=dev, generator=g) * 0.05).contiguous(memory_format=CL)
    y = F.conv2d_fwd(x, w, None, s, p, d)
    dy = torch.randn(y.shape, device=dev, generator=g).contiguous(memory_format=CL)
    dx = F.conv2d_dgrad(dy, F.dgrad_operand(w, x.shape, s, p, d), x.shape, w.shape, s, p, d)
    dw = F.conv2d_wgrad(x, dy, w.shape, s, p, d)
    dot = lambda a, b: float((a.double() * b.double()).sum())
    lhs, via_x, via_w = dot(y, dy), dot(x, dx), dot(w, dw)
    scale = float(y.double().norm() * dy.double().norm())
    assert abs(lhs - via_x) <= 5e-8 * scale, (lhs, via_x, scale)      # fp32 noise ~1e-9 * scale; one wrong 64x64 tile ~3e-5 * scale
    assert abs(lhs - via_w) <= 5e-8 * scale, (lhs, via_w, scale)


@pytest.mark.parametrize("geom", [(4, 128, 1, 128, 128), (4, 3, 64, 128, 128), (2, 21, 64, 192, 176), (4, 20, 64, 128, 128),
                                  (4, 64, 3, 128, 128), (3, 64, 1, 150, 147)])
def test_thin_1x1_weight_gradient(geom, F, dev):
    """The few-channel 1x1 convolutions of the PixelDiscriminator (arch/discriminators.py:70-75) at image resolution take the
    streaming weight-gradient kernel (>= 65536 pixels, one side <= 32 channels): parity with torch in fp64, and with
    accumulation into an existing gradient."""
    N, C, K, H, W = geom
    g = torch.Generator().manual_seed(N * 1000 + C + K)
    x = torch.randn(N, C, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(K, C, 1, 1, generator=g, dtype=torch.float64)
    dy = torch.randn(N, K, H, W, generator=g, dtype=torch.float64)
    ref = torch.einsum("nkhw,nchw->kc", dy, x).reshape(K, C, 1, 1)
    xg, dyg = gpu(x, dev), gpu(dy, dev)
    dw = F.conv2d_wgrad(xg, dyg, w.shape, 1, 0, 1)
    assert rel_err(dw, ref) < 2e-5
    acc = gpu(w, dev).clone()
    F.conv2d_wgrad(xg, dyg, w.shape, 1, 0, 1, out=acc, accumulate=True)
    assert rel_err(acc, ref + w) < 2e-5


BF16_CASES = [(2, 256, 33, 33, 256, 3, 1, 2, 2), (2, 256, 17, 19, 1024, 1, 1, 0, 1), (8, 256, 33, 33, 256, 3, 1, 2, 2),
              (2, 64, 32, 32, 128, 3, 2, 1, 1), (4, 512, 33, 33, 512, 3, 1, 4, 4),
              (2, 21, 40, 40, 64, 7, 1, 3, 1), (2, 64, 40, 40, 21, 7, 1, 3, 1), (2, 20, 17, 19, 36, 3, 1, 1, 1)]   # generic / narrow tiles
# (heads with <= 4 output channels run on the 4x4x1 fp32 MFMA tile in either mode: they are staging-bound, not MFMA-bound)


@pytest.mark.parametrize("shape", BF16_CASES)
def test_conv_bf16_contraction_mode(shape, F, dev):
    """sscg_set_conv_precision(1): operands rounded to bfloat16 (RNE), exact products, fp32 accumulation.  Against torch fp64
    convolutions of the bf16-ROUNDED tensors only the accumulation order differs (2e-5); against the unrounded tensors the
    result must sit at bf16's ~1e-2..1e-3, which shows that the mode is actually engaged."""
    N, C, H, W, K, R, s, p, d = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(K, C, R, R, generator=g) * 0.05
    xr, wr = x.bfloat16().double(), w.bfloat16().double()
    xr.requires_grad_(True)
    wr.requires_grad_(True)
    yr = TF.conv2d(xr, wr, None, s, p, d)
    gy = torch.randn(yr.shape, generator=g)
    gyr = gy.bfloat16().double()
    yr.backward(gyr)
    exact = TF.conv2d(x.double(), w.double(), None, s, p, d)
    try:
        F.set_conv_precision("bf16c")
        assert F.get_conv_precision() == "bf16c"
        xg, wg, gyg = gpu(x, dev), gpu(w, dev), gpu(gy, dev)
        y = F.conv2d_fwd(xg, wg, None, s, p, d)
        dx = F.conv2d_dgrad(gyg, F.dgrad_operand(wg, xg.shape, s, p, d), xg.shape, wg.shape, s, p, d)
        dw = F.conv2d_wgrad(xg, gyg, wg.shape, s, p, d)
    finally:
        F.set_conv_precision("f32")
    assert rel_err(y, yr.detach()) < 2e-5
    assert 1e-4 < rel_err(y, exact) < 3e-2
    assert rel_err(dx, xr.grad) < 2e-5
    assert rel_err(dw, wr.grad) < 5e-5



@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("affine", [True, False])
@pytest.mark.parametrize("act", ["relu", "lrelu"])
def test_norm_bwd_mask_recomputed_from_x_equals_mask_from_y(act, affine, dtype, dev):
    """sscg_norm_bwd with y == NULL (no residual joined the forward): the activation mask recomputed as gamma * xhat + beta > 0 is the
    mask the forward's output carries - dx, dgamma, dbeta are bitwise those of the call that reads y."""
    F = load_sub("functional")
    code, slope = (F.ACT_RELU, 0.0) if act == "relu" else (F.ACT_LRELU, 0.2)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(4, 72, 19, 23, generator=g).to(dev).to(dtype).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(4, 72, 19, 23, generator=g).to(dev).to(dtype).contiguous(memory_format=torch.channels_last)
    gamma = (torch.randn(72, generator=g) * 0.5 + 1.0).to(dev) if affine else None
    beta = (torch.randn(72, generator=g) * 0.3).to(dev) if affine else None
    for per in (False, True, 2):
        mean, rstd = F.norm_stats(x, per)
        y = F.norm_apply(x, mean, rstd, gamma, beta, None, per, code, slope)
        outs = []
        for yy in (y, None):
            dg, db = torch.zeros(72, device=dev), torch.zeros(72, device=dev)
            dx, _ = F.norm_bwd(dy, x, yy, mean, rstd, gamma, per, code, slope, True, False, dg if affine else None, db if affine else None, beta=beta)
            outs.append((dx, dg, db))
        assert torch.equal(outs[0][0], outs[1][0])
        assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])


def test_trainable_batchnorm_affine_under_the_arena_optimiser(dev):
    """A BatchNorm whose weight / bias are trained (the ResNet generators with --norm batch under --honour_nets; DeepLab freezes
    them): dgamma / dbeta are written by the backward kernels and added into the optimiser's gradient arena on the parameter's
    side lane - twice accumulated here (two backward passes), against torch's fp64 BatchNorm."""
    F = load_sub("functional")
    optim = load_sub("optim")
    ops = load_sub("arch.ops")
    torch.manual_seed(0)
    conv = ops.Conv2d(8, 64, 3, 1, 1, bias=False).to(dev)
    bn = ops.BatchNorm2d(64).to(dev)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(64) + 0.5)
        bn.bias.copy_(torch.randn(64) * 0.2)
    opt = optim.FusedAdam(list(conv.parameters()) + list(bn.parameters()), lr=1e-3)
    opt.zero_grad()
    x = torch.randn(4, 8, 13, 11)
    gy = torch.randn(4, 64, 13, 11)
    for _ in range(2):
        y = ops.conv_norm_act(conv, bn, gpu(x, dev), F.ACT_RELU)
        F.backward((y * gpu(gy, dev)).sum())
    F.SideStream.join(dev)
    torch.cuda.synchronize()
    wr = conv.weight.detach().double().cpu().requires_grad_(True)
    gr = bn.weight.detach().double().cpu().requires_grad_(True)
    br = bn.bias.detach().double().cpu().requires_grad_(True)
    yr = torch.relu(TF.batch_norm(TF.conv2d(x.double(), wr, None, 1, 1), None, None, gr, br, True, 0.1, 1e-5))
    (yr * gy.double()).sum().backward()
    assert rel_err(bn.weight._sscg_grad, 2 * gr.grad) < 2e-5
    assert rel_err(bn.bias._sscg_grad, 2 * br.grad) < 2e-5
    assert rel_err(conv.weight._sscg_grad, 2 * wr.grad) < 2e-5


@pytest.mark.parametrize("norm", ["batch", "instance"])
def test_one_node_unit_equals_the_two_node_path_bitwise(norm, dev):
    """arch.ops.conv_norm_act as one autograd node (ConvNormActFn) launches the kernels of Conv2dFn + NormActFn: output, input
    gradient, parameter gradients and BatchNorm state are bitwise those of the two-node path."""
    F = load_sub("functional")
    ops = load_sub("arch.ops")
    torch.manual_seed(1)
    conv = ops.Conv2d(16, 64, 3, 1, 1, bias=(norm == "instance")).to(dev)
    nl = ops.BatchNorm2d(64).to(dev) if norm == "batch" else ops.InstanceNorm2d(64).to(dev)
    x0 = torch.randn(3, 16, 17, 13)
    r0 = torch.randn(3, 64, 17, 13)
    gy = gpu(torch.randn(3, 64, 17, 13), dev)
    outs = []
    for one in (True, False):
        ops.ONE_NODE[0] = one
        try:
            if norm == "batch":
                nl.running_mean.zero_(); nl.running_var.fill_(1.0)
            for p in list(conv.parameters()) + list(nl.parameters()):
                p.grad = None
            x, r = gpu(x0, dev).requires_grad_(True), gpu(r0, dev).requires_grad_(True)
            y = ops.conv_norm_act(conv, nl, x, F.ACT_RELU, residual=r)
            F.backward((y * gy).sum())
            F.SideStream.join(dev)
            torch.cuda.synchronize()
            outs.append([y.detach().clone(), x.grad.clone(), r.grad.clone(), conv.weight.grad.clone()] +
                        ([nl.weight.grad.clone(), nl.bias.grad.clone(), nl.running_mean.clone(), nl.running_var.clone()] if norm == "batch"
                         else [conv.bias.grad.clone()]))
        finally:
            ops.ONE_NODE[0] = True
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_maxpool2x2_and_mse_between_tensors(F, dev):
    """sscg_maxpool2x2_* (torchvision VGG16's MaxPool2d(2, 2), odd sizes floor) and sscg_mse_* (nn.MSELoss between two tensors)."""
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 24, 13, 18, generator=g, dtype=torch.float64)
    xr = x.clone().requires_grad_(True)
    yr = TF.max_pool2d(xr, 2, 2)
    gy = torch.randn(yr.shape, generator=g, dtype=torch.float64)
    yr.backward(gy)
    xg = gpu(x, dev).requires_grad_(True)
    yg = F.maxpool2x2(xg)
    assert rel_err(yg, yr) < 1e-7          # (the fp32 rounding of the fp64 input)
    yg.backward(gpu(gy, dev))
    assert rel_err(xg.grad, xr.grad) < 1e-7
    a = torch.randn(2, 8, 9, 7, generator=g, dtype=torch.float64)
    b = torch.randn(2, 8, 9, 7, generator=g, dtype=torch.float64)
    ar, br = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    lr = TF.mse_loss(ar, br)
    (3.0 * lr).backward()
    ag, bg = gpu(a, dev).requires_grad_(True), gpu(b, dev).requires_grad_(True)
    lg = F.mse_loss(ag, bg)
    assert rel_err(lg, lr) < 1e-6
    F.weighted_sum([lg], [3.0]).backward()
    assert rel_err(ag.grad, ar.grad) < 1e-6 and rel_err(bg.grad, br.grad) < 1e-6


def test_perceptual_loss_vs_reference_golden(dev):
    """utils.perceptual_loss (commented out at model.py:454,462): VGG16 to relu2_2 on both images, MSE between the
    features - against the loss and the gradient the REFERENCE's own utils.perceptual_loss produced on keyed VGG weights
    (g6_perceptual.npz, gen_golden.py g6_perceptual: fp64), and against oracle.nets.perceptual_loss."""
    import os
    from oracle import fixtures as FX
    from oracle import nets
    utils = load_sub("utils")
    F = load_sub("functional")
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "g6_perceptual.npz"))
    vgg = utils.Vgg16(requires_grad=False, weights={"features." + k.split(".", 1)[1]: v for k, v in FX.vgg_state_dict().items()}).to(dev)
    sd = {k: v.detach().double().cpu() for k, v in vgg.state_dict().items()}
    assert list(sd.keys())[:4] == ["slice1.0.weight", "slice1.0.bias", "slice1.2.weight", "slice1.2.bias"] and "slice4.21.bias" in sd
    x, y = FX.vgg_images(torch.float64)
    xr = x.clone().requires_grad_(True)
    lr = nets.perceptual_loss(FX.vgg_state_dict(torch.float64), xr, y)
    lr.backward()
    assert abs(float(lr) - float(gold["loss/f64"])) <= 1e-12 * float(gold["loss/f64"])
    xg = gpu(x, dev).requires_grad_(True)
    lg = utils.perceptual_loss(xg, gpu(y, dev), [0], vgg)
    assert abs(float(lg) - float(gold["loss/f64"])) < 1e-5 * float(gold["loss/f64"])
    lg.backward()
    assert rel_err(F.to_nchw(xg.grad), torch.from_numpy(gold["dx/f64"])) < 1e-4
    out = vgg(gpu(x, dev))
    assert tuple(out["relu4_3"].shape) == (2, 512, 3, 5)


def _sampled_fwd_ref(x, w, idx, s, p, d):
    """fp64 values of conv(x, w) at the sampled output positions idx = (n, k, oy, ox) (gathered receptive fields, zero padding)."""
    N, C, H, W = x.shape
    K, _, R, S = w.shape
    n, k, oy, ox = idx
    out = torch.zeros(n.numel(), dtype=torch.float64, device=x.device)
    xd = x.double()
    wd = w.double()
    for r in range(R):
        for q in range(S):
            iy, ix = oy * s - p + r * d, ox * s - p + q * d
            ok = (iy >= 0) & (iy < H) & (ix >= 0) & (ix < W)
            patch = xd[n, :, iy.clamp(0, H - 1), ix.clamp(0, W - 1)] * ok[:, None]       # [samples, C]
            out += (patch * wd[k, :, r, q]).sum(1)
    return out


def _sampled_dgrad_ref(dy, w, idx, xshape, s, p, d):
    """fp64 values of the data gradient at the sampled input positions idx = (n, c, iy, ix)."""
    N, K, P, Q = dy.shape
    _, C, R, S = w.shape
    n, c, iy, ix = idx
    out = torch.zeros(n.numel(), dtype=torch.float64, device=dy.device)
    dyd = dy.double()
    wd = w.double()
    for r in range(R):
        for q in range(S):
            ty, tx = iy + p - r * d, ix + p - q * d
            oy, ox = torch.div(ty, s, rounding_mode="floor"), torch.div(tx, s, rounding_mode="floor")
            ok = (ty >= 0) & (tx >= 0) & (oy * s == ty) & (ox * s == tx) & (oy < P) & (ox < Q)
            g = dyd[n, :, oy.clamp(0, P - 1), ox.clamp(0, Q - 1)] * ok[:, None]          # [samples, K]
            out += (g * wd[:, c, r, q].t()).sum(1)
    return out


def _sampled_wgrad_ref(x, dy, idx, s, p, d):
    """fp64 values of the weight gradient at the sampled entries idx = (k, c, r, q): sum over every output pixel."""
    N, C, H, W = x.shape
    _, K, P, Q = dy.shape
    k, c, r, q = idx
    out = torch.zeros(k.numel(), dtype=torch.float64, device=x.device)
    for i in range(k.numel()):
        xp = TF.pad(x[:, int(c[i])].double(), (p, p, p, p))
        r0, q0 = int(r[i]) * d, int(q[i]) * d
        win = xp[:, r0:r0 + s * (P - 1) + 1:s, q0:q0 + s * (Q - 1) + 1:s]
        out[i] = (win * dy[:, int(k[i])].double()).sum()
    return out


@pytest.mark.parametrize("shape", _bench_shapes(), ids=lambda s: "%dx%dx%d_c%d_k%d_r%d_s%d_p%d_d%d" % s)
def test_split_contraction_is_fp32_accurate_at_bench_size(shape, F, dev):
    """The split contraction (conv_split.hip: every fp32 operand as three bfloat16 pieces, six exact piece products per pair on the
    bf16 matrix cores, fp32 accumulation) on EVERY convolution shape of the BASELINE step at full size (list recorded by bench.py).
    Ground truth: 4096 sampled outputs of the forward and of the data gradient, each recomputed in fp64 from its receptive field,
    and 96 sampled entries of the weight gradient (both operands split on the fly), each summed in fp64 over every output pixel.
    The split result must be as close to it as the exact-fp32 MFMA kernel is (rms error; both are a few 1e-7 of the tensor's rms) -
    three orders of magnitude below a bf16-rounded contraction - and the adjoint identity must hold between the two split products.
    Shapes the split kernels do not serve (few-channel stems / heads) run the exact kernel in either mode: both errors coincide."""
    N, H, W, C, K, R, s, p, d = shape
    g = torch.Generator(device=dev).manual_seed(sum(shape) + 7)
    x = torch.randn(N, C, H, W, device=dev, generator=g).contiguous(memory_format=CL)
    w = (torch.randn(K, C, R, R, device=dev, generator=g) * (1.0 / (C * R * R) ** 0.5)).contiguous(memory_format=CL)
    P, Q = F.conv_out_size(H, R, s, p, d), F.conv_out_size(W, R, s, p, d)
    dy = torch.randn(N, K, P, Q, device=dev, generator=g).contiguous(memory_format=CL)
    ns = 4096
    ri = lambda hi: torch.randint(0, hi, (ns,), device=dev, generator=g)
    fidx = (ri(N), ri(K), ri(P), ri(Q))
    didx = (ri(N), ri(C), ri(H), ri(W))
    nw = 96
    rw = lambda hi: torch.randint(0, hi, (nw,), device=dev, generator=g)
    widx = (rw(K), rw(C), rw(R), rw(R))
    yref = _sampled_fwd_ref(x, w, fidx, s, p, d)
    dxref = _sampled_dgrad_ref(dy, w, didx, x.shape, s, p, d)
    dwref = _sampled_wgrad_ref(x, dy, widx, s, p, d)
    rms = lambda t: float(t.double().pow(2).mean().sqrt())
    errs = {}
    served = False
    try:
        for mode in ("f32x", "f32s"):
            F.set_conv_precision(mode)
            y = F.conv2d_fwd(x, w, None, s, p, d)
            dx = F.conv2d_dgrad(dy, F.dgrad_operand(w, x.shape, s, p, d), x.shape, w.shape, s, p, d)
            dw = F.conv2d_wgrad(x, dy, w.shape, s, p, d)
            errs[mode] = (rms(y[fidx].double() - yref) / rms(yref), rms(dx[didx].double() - dxref) / rms(dxref),
                          rms(dw[widx].double() - dwref) / rms(dwref))
            if mode == "f32s":
                served = F.split_applies(x.shape, w.shape, s, p, d, 0, 0) or F.split_applies(x.shape, w.shape, s, p, d, 0, 1)
                dot = lambda a, b: float((a.double() * b.double()).sum())
                lhs, via_x, via_w = dot(y, dy), dot(x, dx), dot(w, dw)
                assert abs(lhs - via_x) <= 5e-8 * float(y.double().norm() * dy.double().norm()), (lhs, via_x)
                assert abs(lhs - via_w) <= 5e-8 * float(y.double().norm() * dy.double().norm()), (lhs, via_w)
    finally:
        F.set_conv_precision("f32")
    print("rms error against fp64 samples (fwd, dgrad, wgrad): exact fp32 %s, split %s, split kernels used: %s" % (errs["f32x"], errs["f32s"], served))
    for e_split, e_exact in zip(errs["f32s"][:2], errs["f32x"][:2]):
        assert e_split < 2e-6
        assert e_split <= 1.1 * e_exact + 2e-8
    # weight gradient: the reduction runs over up to 524288 pixels (cut into chunks of a few thousand per workgroup); once the
    # running sum is ~100x a single product, the smallest piece products (2^-16 of a product) fall below its last bit, so on the
    # longest reductions the split result is up to 2x the exact kernel's error - still fp32 roundoff class (1.4e-6 at worst
    # against 2.2e-6 for the exact kernel's own worst shape); 96 samples: +-10 % on either estimate
    assert errs["f32s"][2] < 3e-6
    assert errs["f32s"][2] <= 2.2 * errs["f32x"][2] + 5e-8



_WGRAD_SPLIT_CASES = [
    # N, H, W, C, K, R, stride, pad, dil  - ragged on purpose: K, C*R*S and the pixel count are no multiples of the 128x128x16 tile
    (1, 37, 29, 72, 136, 3, 1, 1, 1),
    (2, 33, 47, 128, 256, 3, 2, 1, 1),
    (1, 40, 40, 264, 128, 3, 1, 2, 2),
    (2, 32, 32, 256, 512, 1, 1, 0, 1),
    (3, 19, 23, 16, 200, 4, 2, 1, 1),
]


@pytest.mark.parametrize("shape", _WGRAD_SPLIT_CASES, ids=lambda s: "%dx%dx%d_c%d_k%d_r%d_s%d_p%d_d%d" % s)
@pytest.mark.parametrize("variant", ["plan", "fly", "planes", "planes_2stage", "planes_split3", "fly_64x64"])
def test_split_weight_gradient_variants(shape, variant, F, dev):
    """Every route of the fp32-accurate weight gradient (precision 2): the library's own plan, the on-the-fly split in
    conv_wgrad.hip (both tile classes), and wgrads_kernel of conv_split.hip (operands pre-split into bf16 planes; three and two
    copy stages, forced pixel splits, 1x1 filters) against the fp64 weight gradient of torch - full tensor, ragged tiles."""
    N, H, W, C, K, R, s, p, d = shape
    g = torch.Generator(device=dev).manual_seed(sum(shape) + 11)
    x = torch.randn(N, C, H, W, device=dev, generator=g).contiguous(memory_format=CL)
    P, Q = F.conv_out_size(H, R, s, p, d), F.conv_out_size(W, R, s, p, d)
    dy = torch.randn(N, K, P, Q, device=dev, generator=g).contiguous(memory_format=CL)
    ref = torch.nn.grad.conv2d_weight(x.double(), (K, C, R, R), dy.double(), stride=s, padding=p, dilation=d)
    kw = {"plan": {}, "fly": dict(wgrad_class=2), "planes": dict(wgrad_class=3), "planes_2stage": dict(wgrad_class=3, wgrad_flags=1),
          "planes_split3": dict(wgrad_class=3, wgrad_splits=3), "fly_64x64": dict(wgrad_class=1, wgrad_splits=2)}[variant]
    old = F.tuning(**kw)
    try:
        F.set_conv_precision("f32s")
        dw = F.conv2d_wgrad(x, dy, (K, C, R, R), s, p, d)
        acc = torch.ones_like(dw)
        F.conv2d_wgrad(x, dy, (K, C, R, R), s, p, d, out=acc, accumulate=True)
    finally:
        F.TUNING[0], F.WGRAD_TUNING[0] = old
        F.set_conv_precision("f32")
    err = float((dw.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    assert err < 1e-6, err
    assert float((acc.double() - 1 - ref).abs().max()) <= 2e-6 * float(ref.abs().max()) + 1e-6


def _torch_pixel_discriminator(net, norm, double=True):
    """The reference's PixelDiscriminator (arch/discriminators.py:66-80) as stock torch modules holding `net`'s weights."""
    from torch import nn
    convs = [m for m in net.dis_model if hasattr(m, "kernel_size")]
    c1, c2, c3 = convs
    nl = nn.BatchNorm2d(c2.out_channels) if norm == "batch" else nn.InstanceNorm2d(c2.out_channels)
    ref = nn.Sequential(nn.Conv2d(c1.in_channels, c1.out_channels, 1), nn.LeakyReLU(0.2), nn.Conv2d(c2.in_channels, c2.out_channels, 1, bias=c2.bias is not None),
                        nl, nn.LeakyReLU(0.2), nn.Conv2d(c3.in_channels, 1, 1, bias=c3.bias is not None))
    with torch.no_grad():
        for r, m in zip((ref[0], ref[2], ref[5]), convs):
            r.weight.copy_(m.weight.detach().cpu())
            if m.bias is not None:
                r.bias.copy_(m.bias.detach().cpu())
        if norm == "batch":
            ours = [m for m in net.dis_model if getattr(m, "running_mean", None) is not None][0]
            nl.weight.copy_(ours.weight.detach().cpu())
            nl.bias.copy_(ours.bias.detach().cpu())
    return ref.double() if double else ref


@pytest.mark.parametrize("norm", ["instance", "batch"])
@pytest.mark.parametrize("geom", [(3, 3, 64, 37, 29), (2, 21, 64, 64, 48), (2, 3, 32, 40, 40), (1, 4, 8, 33, 17)],
                         ids=lambda g: "n%d_c%d_ndf%d_%dx%d" % g)
def test_fused_pixel_discriminator_tail(geom, norm, F, dev):
    """PixelDiscriminator with its tail (norm -> LeakyReLU -> Conv2d(2 ndf, 1, 1x1)) formed in one pass over the 2 ndf-channel map
    (ConvNormActHeadFn: sscg_norm_head_fwd / sscg_norm_head_bwd) against the reference's module in fp64: output, input gradient,
    every parameter gradient; and against the unfused launch sequence of the same library (SSCG_FUSE_HEAD=0 path).  ndf = 8 (16
    channels) is the smallest width the fused pass serves."""
    ops = load_sub("arch.ops")
    disc = load_sub("arch.discriminators")
    N, Cin, ndf, H, W = geom
    torch.manual_seed(sum(geom))
    nl = ops.get_norm_layer(norm)
    net = disc.PixelDiscriminator(Cin, ndf, norm_layer=nl, use_bias=(norm == "instance")).to(dev)
    with torch.no_grad():
        for p in net.parameters():
            p.copy_(torch.randn(p.shape) * (0.5 if p.dim() == 1 else 1.0 / p.shape[1] ** 0.5))
    ref = _torch_pixel_discriminator(net, norm)
    x0 = torch.randn(N, Cin, H, W)
    g0 = torch.randn(N, 1, H, W)
    xr = x0.double().requires_grad_(True)
    yr = ref(xr)
    (yr * g0.double()).sum().backward()
    want = [yr, xr.grad] + [p.grad for p in ref.parameters() if p.grad is not None]
    got = {}
    for fused in (True, False):
        ops.FUSE_HEAD[0] = fused
        try:
            for p in net.parameters():
                p.grad = None
            x = gpu(x0, dev).requires_grad_(True)
            y = net(x)
            F.backward((y * gpu(g0, dev)).sum())
            F.SideStream.join(dev)
            torch.cuda.synchronize()
            convs = [m for m in net.dis_model if hasattr(m, "kernel_size")]
            norms = [m for m in net.dis_model if isinstance(m, ops.BatchNorm2d)]
            grads = []
            for m in convs:
                grads.append(m.weight.grad)
                if m.bias is not None:
                    grads.append(m.bias.grad)
            got[fused] = [y.detach(), x.grad] + grads + ([norms[0].weight.grad, norms[0].bias.grad] if norms else [])
        finally:
            ops.FUSE_HEAD[0] = True
    # the reference's parameter order: conv1 (w, b), conv2 (w[, b]), norm (w, b) for batch, conv3 (w[, b])
    names = ["y", "dx", "w1", "b1", "w2"] + (["b2"] if norm == "instance" else []) + (["gamma", "beta"] if norm == "batch" else []) + ["w3"] + (["b3"] if norm == "instance" else [])
    refs = dict(zip(names, want))
    ours_names = ["y", "dx", "w1", "b1", "w2"] + (["b2"] if norm == "instance" else []) + ["w3"] + (["b3"] if norm == "instance" else []) + (["gamma", "beta"] if norm == "batch" else [])
    for fused in (True, False):
        vals = dict(zip(ours_names, got[fused]))
        for k in names:
            if k == "b2":
                continue        # a bias in front of InstanceNorm has zero gradient: nothing to compare against but roundoff
            e = rel_err(vals[k].reshape(refs[k].shape), refs[k])
            assert e < 3e-5, (fused, k, e)
    a, b = dict(zip(ours_names, got[True])), dict(zip(ours_names, got[False]))
    for k in ("y", "dx", "w2", "w3"):
        assert rel_err(a[k], b[k]) < 1e-5, k


@pytest.mark.parametrize("geom", [(3, 3, 37, 29, "instance", None), (2, 21, 64, 48, "instance", None), (2, 20, 33, 65, "batch", None),
                                  (1, 4, 40, 40, "instance", 0), (2, 21, 64, 64, "instance", 0), (3, 3, 50, 70, "batch", 0),
                                  (8, 3, 256, 256, "instance", None), (8, 21, 256, 256, "instance", None)],
                         ids=lambda g: "n%d_c%d_%dx%d_%s_class%s" % g)
def test_fused_pixel_discriminator_front(geom, F, dev):
    """PixelDiscriminator's front half as one launch (PixelDiscFn / sscg_conv2d_front_fwd, arch/discriminators.py:70-73): Conv2d(cin, 64,
    1x1) -> LeakyReLU formed per workgroup in LDS in front of the split contraction of Conv2d(64, 128, 1x1), whose epilogue takes the
    norm layer's statistics.  Against the reference's module in fp64 - output, input gradient, every parameter gradient - and against
    the unfused launch sequence of the same library (FUSE_FRONT off): 64x64 tiles (small maps), 128x128 tiles (forced, and chosen at
    the bench size 8 x 256 x 256), both norm kinds; 3 / 4 / 20 / 21 input channels."""
    ops = load_sub("arch.ops")
    disc = load_sub("arch.discriminators")
    N, Cin, H, W, norm, cls = geom
    if F.get_conv_precision() != "f32s":
        pytest.skip("the fused front half lives in the split contraction")
    torch.manual_seed(N + Cin + H + W)
    nl = ops.get_norm_layer(norm)
    net = disc.PixelDiscriminator(Cin, 64, norm_layer=nl, use_bias=(norm == "instance")).to(dev)
    with torch.no_grad():
        for p in net.parameters():
            p.copy_(torch.randn(p.shape) * (0.5 if p.dim() == 1 else 1.0 / p.shape[1] ** 0.5))
    ref = _torch_pixel_discriminator(net, norm)
    x0 = torch.randn(N, Cin, H, W)
    g0 = torch.randn(N, 1, H, W)
    xr = x0.double().requires_grad_(True)
    torch.set_num_threads(min(torch.get_num_threads(), 32))
    yr = ref(xr)
    (yr * g0.double()).sum().backward()
    want = [yr, xr.grad] + [p.grad for p in ref.parameters() if p.grad is not None]
    got = {}
    old = F.tuning(tile_class=cls)
    calls = []
    try:
        for fused in (True, False):
            F.FUSE_FRONT[0] = fused
            for p in net.parameters():
                p.grad = None
            x = gpu(x0, dev).requires_grad_(True)
            n0 = F.FRONT_CALLS[0]
            y = net(x)
            calls.append(F.FRONT_CALLS[0] - n0)
            F.backward((y * gpu(g0, dev)).sum())
            F.SideStream.join(dev)
            torch.cuda.synchronize()
            convs = [m for m in net.dis_model if hasattr(m, "kernel_size")]
            norms = [m for m in net.dis_model if isinstance(m, ops.BatchNorm2d)]
            grads = []
            for m in convs:
                grads.append(m.weight.grad)
                if m.bias is not None:
                    grads.append(m.bias.grad)
            got[fused] = [y.detach(), x.grad] + grads + ([norms[0].weight.grad, norms[0].bias.grad] if norms else [])
            # without a backward pass nothing but y is written: same output
            with torch.no_grad():
                y2 = net(gpu(x0, dev))
            assert torch.equal(y2, y.detach()), fused
    finally:
        F.FUSE_FRONT[0] = True
        F.TUNING[0], F.WGRAD_TUNING[0] = old
    assert calls == [1, 0], calls          # the fused launch ran exactly where it was asked for
    names = ["y", "dx", "w1", "b1", "w2"] + (["b2"] if norm == "instance" else []) + (["gamma", "beta"] if norm == "batch" else []) + ["w3"] + (["b3"] if norm == "instance" else [])
    refs = dict(zip(names, want))
    ours_names = ["y", "dx", "w1", "b1", "w2"] + (["b2"] if norm == "instance" else []) + ["w3"] + (["b3"] if norm == "instance" else []) + (["gamma", "beta"] if norm == "batch" else [])
    def l2(a_, b_):
        a_, b_ = a_.detach().double().cpu().reshape(-1), b_.detach().double().cpu().reshape(-1)
        return float((a_ - b_).norm() / b_.norm().clamp_min(1e-30))
    errs, errs2 = {}, {}
    for fused in (True, False):
        vals = dict(zip(ours_names, got[fused]))
        for k in names:
            if k == "b2":
                continue        # a bias in front of InstanceNorm has zero gradient: nothing to compare against but roundoff
            errs[(fused, k)] = rel_err(vals[k].reshape(refs[k].shape), refs[k])
            errs2[(fused, k)] = l2(vals[k].reshape(refs[k].shape), refs[k])
    print()
    print("  ".join("%s %.1e/%.1e" % (k, errs[(True, k)], errs[(False, k)]) for k in names if k != "b2"), "(max norm, fused / unfused, against fp64)")
    print("  ".join("%s %.1e/%.1e" % (k, errs2[(True, k)], errs2[(False, k)]) for k in names if k != "b2"), "(rel-L2)")
    big = N * H * W >= 1 << 18
    for k in names:
        if k == "b2":
            continue
        if not big:
            assert errs[(True, k)] < 3e-5 and errs[(False, k)] < 3e-5, (k, errs[(True, k)], errs[(False, k)])
            continue
        # bench size (524288 pixels, InstanceNorm over 65536): the forward holds 3e-5 in the max norm.  The gradients pass two LeakyReLU
        # masks - of ~1e8 pre-activations a few lie within fp32 rounding of zero and flip against fp64, each moving a handful of
        # gradient elements by percents of the tensor's maximum (tests/test_nets_gpu.py: the flip metric) - and sum 524288 pixels in
        # fp32: held in rel-L2, the max norm confined to the flip bound, and never worse than the established unfused sequence
        # by more than its own distance to fp64
        if k == "y":
            assert errs[(True, k)] < 3e-5 and errs[(False, k)] < 3e-5, (k, errs[(True, k)], errs[(False, k)])
        else:
            assert errs2[(True, k)] < max(1e-4, 4 * errs2[(False, k)]) and errs[(True, k)] < 5e-2, (k, errs2[(True, k)], errs2[(False, k)], errs[(True, k)])
            assert errs2[(False, k)] < 1e-3, (k, errs2[(False, k)])
    a, b = dict(zip(ours_names, got[True])), dict(zip(ours_names, got[False]))
    for k in ("y", "dx", "w1", "w2", "w3"):
        assert (l2(a[k], b[k]) if big else rel_err(a[k], b[k])) < (1e-3 if (big and k != "y") else 1e-5), k


@pytest.mark.parametrize("case", [("instance", 3, 64, 20, 24, 128, 3, 1, 1), ("batch", 2, 128, 33, 33, 128, 1, 0, 1), ("batch2", 4, 64, 17, 19, 256, 3, 2, 2),
                                  ("instance", 2, 256, 33, 33, 64, 1, 0, 1), ("batch", 8, 32, 16, 16, 32, 3, 1, 1)],
                         ids=lambda c: "%s_n%d_c%d_%dx%d_k%d_r%d_p%d_d%d" % c)
@pytest.mark.parametrize("act", ["relu", "lrelu", "none"])
def test_norm_backward_sums_fused_into_the_data_gradient(case, act, dev):
    """conv_a -> norm -> activation -> conv_b: conv_b's data gradient IS the upstream gradient of the normalisation layer, and its
    epilogue takes that layer's backward sums (sscg_conv2d_dgrad_bsums + sscg_norm_bwd_from_sums) - the reduction pass over (dz, y)
    does not run.  Same fp64 sums in another order: input gradient, the norm layer's weight / bias gradients and conv_a's weight
    gradient agree with the unfused path to 1e-5 of the tensor's scale (tiles inside one group sum four rows in fp32 first), over ragged tiles, tiles that straddle a group boundary
    (InstanceNorm images, stacked BatchNorm groups) and both tile classes."""
    F = load_sub("functional")
    ops = load_sub("arch.ops")
    arch = load_sub("arch")
    norm, n, c, h, w, k, r, pad, dil = case
    torch.manual_seed(5)
    conv_a = ops.Conv2d(32, c, 3, 1, 1, bias=(norm == "instance")).to(dev)
    nl = ops.InstanceNorm2d(c).to(dev) if norm == "instance" else ops.BatchNorm2d(c).to(dev)
    conv_b = ops.Conv2d(c, k, r, 1, pad, dilation=dil, bias=False).to(dev)
    a = {"relu": F.ACT_RELU, "lrelu": F.ACT_LRELU, "none": F.ACT_NONE}[act]
    x0 = torch.randn(n, 32, h, w)
    gy = None
    outs, used = [], []
    real, was = F.norm_bwd_from_sums, F.FUSE_BSUMS[0]
    for fused in (True, False):
        F.FUSE_BSUMS[0] = fused
        calls = []
        F.norm_bwd_from_sums = lambda *aa, **kk: (calls.append(1), real(*aa, **kk))[1]
        try:
            for p in list(conv_a.parameters()) + list(nl.parameters()) + list(conv_b.parameters()):
                p.grad = None
            if norm != "instance":
                nl.running_mean.zero_(); nl.running_var.fill_(1.0)
            x = gpu(x0, dev).requires_grad_(True)
            with arch.batch_groups(2 if norm == "batch2" else 1):
                z = ops.conv_norm_act(conv_a, nl, x, a, slope=0.2)
            out = conv_b(z)
            if gy is None:
                gy = gpu(torch.randn(out.shape), dev)
            F.backward((out * gy).sum())
            F.SideStream.join(dev)
            torch.cuda.synchronize()
            # (conv_a's bias gradient under InstanceNorm is zero in exact arithmetic - both routes return rounding noise: not compared)
            outs.append([x.grad.clone(), conv_a.weight.grad.clone()] + ([nl.weight.grad.clone(), nl.bias.grad.clone()] if norm != "instance" else []))
            used.append(len(calls))
        finally:
            F.FUSE_BSUMS[0] = was
            F.norm_bwd_from_sums = real
    assert used == [1, 0], used          # the fused route was taken exactly when it was on
    for t_f, t_u in zip(*outs):
        assert float((t_f.double() - t_u.double()).abs().max()) <= 1e-5 * float(t_u.double().abs().max()) + 1e-9


@pytest.mark.parametrize("geom", [(2, 64, 33, 33, 1), (8, 32, 17, 19, 1), (2, 64, 16, 16, 2)], ids=lambda g: "n%d_p%d_%dx%d_groups%d" % g)
def test_residual_fan_in_joins_in_the_data_gradient(geom, dev):
    """Bottleneck chain (arch/generators.py:345-365): a block's input feeds conv1 and the shortcut.  The shortcut's gradient (the masked
    gradient the bn3 + residual -> ReLU unit leaves) joins in conv1's data gradient (sscg_conv2d_dgrad_bsums / _add with `addend`) - no
    add pass - and that launch also takes the backward sums of the PREVIOUS block's bn3 unit, its mask read off the unit's output
    (`nz`) - no reduction pass.  Input gradient and every weight gradient agree with the separate passes to 1e-5 of the tensor's scale."""
    F = load_sub("functional")
    gen = load_sub("arch.generators")
    arch = load_sub("arch")
    n, planes, h, w, groups = geom
    torch.manual_seed(11)
    blocks = [gen.Bottleneck(4 * planes, planes).to(dev) for _ in range(3)]
    x0 = torch.randn(n, 4 * planes, h, w)
    gy = None
    outs, adds, sums = [], [], []
    real_sums, real_add, was = F.norm_bwd_from_sums, F.add, (F.FUSE_JOIN[0], F.FUSE_BSUMS[0])
    for fused in (True, False):
        F.FUSE_JOIN[0], F.FUSE_BSUMS[0] = fused, True
        c_sums, c_add = [], []
        F.norm_bwd_from_sums = lambda *aa, **kk: (c_sums.append(1), real_sums(*aa, **kk))[1]
        F.add = lambda *aa, **kk: (c_add.append(1), real_add(*aa, **kk))[1]
        try:
            params = [p for b in blocks for p in b.parameters() if p.requires_grad]
            for p in params:
                p.grad = None
            for b in blocks:
                for bn in (b.bn1, b.bn2, b.bn3):
                    bn.running_mean.zero_(); bn.running_var.fill_(1.0)
            x = gpu(x0, dev).requires_grad_(True)
            y = x
            with arch.batch_groups(groups):
                for b in blocks:
                    y = b(y)
            if gy is None:
                gy = gpu(torch.randn(y.shape), dev)
            F.backward((y * gy).sum())
            F.SideStream.join(dev)
            torch.cuda.synchronize()
            outs.append([x.grad.clone()] + [p.grad.clone() for p in params])
            adds.append(len(c_add)); sums.append(len(c_sums))
        finally:
            F.FUSE_JOIN[0], F.FUSE_BSUMS[0] = was
            F.norm_bwd_from_sums, F.add = real_sums, real_add
    assert adds == [0, 3], adds          # three fan-ins: joined in conv1's data gradient / three add passes
    assert sums == [8, 6], sums          # bn1, bn2 of every block either way; bn3 of blocks 1 and 2 only when their consumer joins the fan-in
    for t_f, t_u in zip(*outs):
        assert float((t_f.double() - t_u.double()).abs().max()) <= 1e-5 * float(t_u.double().abs().max()) + 1e-9


def test_operand_copies_of_many_weights_in_one_launch_are_bit_identical(F, dev):
    """sscg_weight_krsc_to_crsk_batch (one launch for every stale transposed operand copy of a step) against the per-weight
    sscg_weight_krsc_to_crsk: the three destination forms, ragged channel counts, and the in-place rewrite of the second refresh."""
    shapes = [(64, 3, 7, 7), (256, 256, 3, 3), (21, 2048, 3, 3), (1024, 256, 1, 1), (40, 24, 3, 3), (128, 64, 4, 4), (33, 65, 1, 1)]
    kinds = {"t32": torch.float32, "t16": torch.bfloat16, "tx3": "x3"}
    g = torch.Generator().manual_seed(5)
    ws = [gpu(torch.randn(s, generator=g), dev) for s in shapes]
    was, mode = F.BATCH_TRANSPOSES[0], F.get_conv_precision()
    ptrs = None
    try:
        F.set_conv_precision("f32x")        # (weight_bf16 then reads a cached cast of the weight, not an optimiser's shadow arena)
        F.BATCH_TRANSPOSES[0] = True
        for rnd in range(2):
            for w in ws:
                w.mul_(1.25)                # a new version: every copy is stale
            F._IN_REFRESH[0] = True
            try:
                jobs = []
                for w in ws:
                    F._refresh_kinds(w, tuple(kinds), jobs)
                assert len(jobs) == 3 * len(ws)
                F._transpose_batch(jobs)
            finally:
                F._IN_REFRESH[0] = False
            now = [[getattr(w, F._WT_ATTR[k]).t.data_ptr() for k in kinds] for w in ws]
            assert ptrs is None or ptrs == now      # the second refresh rewrote the copies of the first in place
            ptrs = now
            for w in ws:
                for k, dtype in kinds.items():
                    ta, tb = getattr(w, F._WT_ATTR[k]).t, F.weight_transposed(w, dtype)
                    assert ta.dtype == tb.dtype and ta.shape == tb.shape and ta.stride() == tb.stride()
                    bits = torch.int16 if ta.dtype == torch.bfloat16 else torch.int32
                    assert torch.equal(ta.view(bits), tb.view(bits)), (tuple(w.shape), k)
            jobs = []
            F._IN_REFRESH[0] = True
            try:
                for w in ws:
                    F._refresh_kinds(w, tuple(kinds), jobs)
            finally:
                F._IN_REFRESH[0] = False
            assert jobs == []               # everything is fresh now
    finally:
        F.BATCH_TRANSPOSES[0] = was
        F.set_conv_precision(mode)


# ------------------------------------------------------------------------------------------ fused store phases against fp64, bench size
def _dgrad_fp64(dy, w, H, W, pad, dil):
    """fp64 data gradient of a stride-1 convolution on the device: dx[n, c, y, x] = sum_{ky, kx, k} dy[n, k, y + pad - ky*dil, x + pad - kx*dil]
    w[k, c, ky, kx], as R*S matrix products over shifted views (torch has no fp64 convolution on this device)."""
    N, K, P, Q = dy.shape
    _, C, R, S = w.shape
    m = (R - 1) * dil
    dyp = TF.pad(dy.double().permute(0, 2, 3, 1), (0, 0, m, m, m, m))           # [N][P + 2m][Q + 2m][K]
    w64 = w.double()
    dx = torch.zeros(N, H, W, C, dtype=torch.float64, device=dy.device)
    for ky in range(R):
        for kx in range(S):
            oy, ox = m + pad - ky * dil, m + pad - kx * dil
            dx += dyp[:, oy:oy + H, ox:ox + W, :] @ w64[:, :, ky, kx]
    return dx.permute(0, 3, 1, 2)


# (N, H, W, C, K, R, pad, dil, residual unit in front, addend): the DeepLab layer3 Bottleneck's three data gradients at the BASELINE
# step's size (8 x 33 x 33 = 8712 rows: 69 tiles of 128 rows, the last one ragged; /root/reference arch/generators.py:345-365)
_STORE_PHASE_CASES = [
    (8, 33, 33, 1024, 256, 1, 0, 1, True, True),      # conv1: its input is the previous block's relu(bn3 + shortcut); the shortcut's gradient joins
    (8, 33, 33, 256, 1024, 1, 0, 1, False, False),    # conv3: its input is relu(bn2(conv2))
    (8, 33, 33, 256, 256, 3, 2, 2, False, False),     # conv2 (dilation 2): its input is relu(bn1(conv1))
    (16, 33, 33, 1024, 256, 1, 0, 1, True, True),     # the stacked Gsi pass: two BatchNorm groups of 8712 rows in one launch
]


@pytest.mark.parametrize("case", _STORE_PHASE_CASES, ids=lambda c: "%dx%dx%d_c%d_k%d_r%d_p%d_d%d_res%d_add%d" % tuple(int(v) for v in c))
def test_fused_store_phases_against_fp64_at_bench_size(case, F, dev):
    """the data gradient's fused store phases - the fan-in `addend`, the normalisation backward's sums with the mask
    recomputed from nx or read off the unit's output - were only compared with the UNFUSED HIP passes, at small shapes.  Here, at the
    bench step's size, against fp64: (a) dx = dgrad(dy, w) + addend, every element; (b) the backward of the unit in front finished
    from the launch's records (sscg_norm_bwd_from_sums): its input gradient, d gamma, d beta and the shortcut's gradient, against the
    batch-norm backward formula evaluated in fp64 on the fp64 gradient."""
    N, H, W, C, K, R, pad, dil, res, add = case
    groups = 2 if N == 16 else 1
    g = torch.Generator(device=dev).manual_seed(sum(int(v) for v in case) + 3)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g)
    w = (rn(K, C, R, R) * (1.0 / (C * R * R) ** 0.5)).contiguous(memory_format=CL)
    dy = rn(N, K, H, W).contiguous(memory_format=CL)
    nx = (rn(N, C, H, W) * 1.7 + 0.3).contiguous(memory_format=CL)
    gamma, beta = (rn(C) * 0.3 + 1.0), rn(C) * 0.2
    resid = rn(N, C, H, W).contiguous(memory_format=CL) if res else None
    addend = rn(N, C, H, W).contiguous(memory_format=CL) if add else None
    per_sample = False if groups == 1 else groups
    mean, rstd = F.norm_stats(nx, per_sample)
    z = F.norm_apply(nx, mean, rstd, gamma, beta, resid, per_sample, F.ACT_RELU)        # the unit's forward output = this conv's input
    F.set_conv_precision("f32s")
    try:
        wt = F.dgrad_operand(w, z.shape, 1, pad, dil)
        L = (N // groups) * H * W
        info = (nx, mean, rstd, gamma, beta, (groups, L, C), F.ACT_RELU, 0.0) + ((True,) if res else ())
        dx, rec, joined = F.conv2d_dgrad(dy, wt, z.shape, w.shape, 1, pad, dil, bsums=info, addend=addend, z=z)
        assert rec is not None and (joined or not add), "the fused route must serve this shape (records %s, joined %s)" % (rec is not None, joined)
        dgb = torch.empty((2, C), dtype=torch.float32, device=dev)
        dnx, dres = F.norm_bwd_from_sums(rec, dx, nx, mean, rstd, gamma, beta, per_sample, F.ACT_RELU, 0.0, dgb[0], dgb[1],
                                         y=z if res else None, want_dres=res)
    finally:
        F.set_conv_precision("f32")
    torch.cuda.synchronize()
    # ---- fp64
    dx64 = _dgrad_fp64(dy, w, H, W, pad, dil)
    if add:
        dx64 = dx64 + addend.double()
    rms = lambda t: float(t.double().pow(2).mean().sqrt())
    e_dx = rms(dx.double() - dx64) / rms(dx64)
    mask = (z > 0).double()            # the forward's own mask (a fp64 re-evaluation would flip the few elements within rounding of 0)
    gg = mask * dx64
    gv = lambda t: t.double().reshape(groups, N // groups, C, H, W)
    xh = (gv(nx) - mean.double().reshape(groups, 1, C, 1, 1)) * rstd.double().reshape(groups, 1, C, 1, 1)
    s1 = gv(gg).sum((1, 3, 4), keepdim=True)
    s2 = (gv(gg) * xh).sum((1, 3, 4), keepdim=True)
    dnx64 = (gamma.double().reshape(1, 1, C, 1, 1) * rstd.double().reshape(groups, 1, C, 1, 1) * (gv(gg) - s1 / L - xh * s2 / L)).reshape(N, C, H, W)
    e_dnx = rms(dnx.double() - dnx64) / rms(dnx64)
    dgamma64, dbeta64 = s2.sum(0).flatten(), s1.sum(0).flatten()
    scale = float((gv(gg).pow(2).sum((0, 1, 3, 4)).sqrt()).max())          # a sum of L terms: error relative to the L2 norm of its terms
    e_dg = float((dgb[0].double() - dgamma64).abs().max()) / scale
    e_db = float((dgb[1].double() - dbeta64).abs().max()) / scale
    # the store phase's OWN arithmetic, apart from the error dx already carries: the kernel's sums against fp64 sums of the dx it wrote
    ggh = gv(mask * dx.double())
    i_db = float((dgb[1].double() - ggh.sum((1, 3, 4)).sum(0)).abs().max()) / scale
    i_dg = float((dgb[0].double() - (ggh * xh).sum((1, 3, 4)).sum(0)).abs().max()) / scale
    print("dx rms err %.2e, norm-backward input gradient %.2e, d gamma %.2e, d beta %.2e (of the terms' L2 norm); the sums against fp64 sums "
          "of the kernel's own dx: d gamma %.2e, d beta %.2e" % (e_dx, e_dnx, e_dg, e_db, i_dg, i_db))
    assert e_dx < 2e-6
    assert e_dnx < 5e-6
    # (a sum over 8712 rows also collects the part of dx's rounding error that is COHERENT over the rows of a channel - the same weight
    #  pieces meet every row: measured 6-9 x the incoherent estimate e_dx, i.e. a per-element bias of ~5e-8 of the tensor's rms.  The
    #  unfused reduction over the same dx shows the same figure: it is dx's, not the store phase's - which is held to 1e-6 by itself.)
    assert e_dg < 2e-5 and e_db < 2e-5
    assert i_dg < 1e-6 and i_db < 1e-6
    if res:
        assert rms(dres.double() - gg) / rms(gg) < 2e-6


@pytest.mark.parametrize("case", [(8, 33, 33, 1024, 256, 1, 0, 1), (16, 33, 65, 1024, 256, 1, 0, 1), (8, 33, 33, 256, 256, 3, 2, 2)],
                         ids=lambda c: "%dx%dx%d_c%d_k%d_r%d_p%d_d%d" % c)
def test_fused_fan_in_against_fp64_at_bench_size_bf16(case, F, dev):
    """The bf16 twin (conv16_kernel's store phase; configs 3 / 5 sizes): dx = bf16(dgrad(dy, w) + addend) against fp64 on the
    bf16-rounded operands - within the two bf16 roundings of the passes it replaces."""
    N, H, W, C, K, R, pad, dil = case
    g = torch.Generator(device=dev).manual_seed(sum(case) + 5)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g)
    w = (rn(K, C, R, R) * (1.0 / (C * R * R) ** 0.5)).contiguous(memory_format=CL)
    dy = rn(N, K, H, W).contiguous(memory_format=CL).to(torch.bfloat16)
    addend = rn(N, C, H, W).contiguous(memory_format=CL).to(torch.bfloat16)
    wt = F.weight_transposed(w, torch.bfloat16)
    dx, rec, joined = F.conv2d_dgrad(dy, wt, (N, C, H, W), w.shape, 1, pad, dil, out_dtype=torch.bfloat16, addend=addend)
    assert joined and dx.dtype == torch.bfloat16
    torch.cuda.synchronize()
    d64 = _dgrad_fp64(dy.float(), w.to(torch.bfloat16).float(), H, W, pad, dil)
    ref = d64 + addend.double()
    err = (dx.double() - ref).abs()
    # bit-identical to the separate passes it replaces (bf16 tensors in HBM between kernels): dx = bf16(bf16(dgrad) + addend) - two
    # roundings to nearest, each at most 2^-9 of its value (2^-8 of the binade's lower end), + the fp32 accumulation's noise
    excess = float((err - (d64.abs() + ref.abs()) * (2.0 ** -8)).max())
    assert excess <= 2e-5, excess
    assert float(err.pow(2).mean().sqrt()) / float(ref.pow(2).mean().sqrt()) < 4e-3


# ------------------------------------------------------------------------------------------ every tile class of the split family
# (N, C, H, W, K, R, stride, pad, dil): ragged rows, a long reduction (tail split-K), a short one, stride 2 (parity-class data gradient)

_KS_CLASS_CASES = [(2, 256, 33, 33, 256, 3, 1, 2, 2), (3, 96, 19, 23, 160, 3, 1, 1, 1), (2, 32, 17, 17, 64, 1, 1, 0, 1), (2, 1024, 17, 17, 256, 1, 1, 0, 1),
                   (2, 64, 32, 32, 128, 3, 2, 1, 1),
                   # heads: 21 / 20 output channels (always the 128x32 class, whatever is forced): the DeepLab classifier (every tile cut
                   # along K, element-wise reduce), the ResNet generators' 7x7 head, a 20-channel one (16-byte row segments)
                   (2, 2048, 9, 9, 21, 3, 1, 6, 6), (2, 64, 24, 24, 21, 7, 1, 3, 1), (2, 64, 17, 19, 20, 3, 1, 1, 1)]


@pytest.mark.parametrize("cls", [0, 1, 2, 3], ids=["128x128", "64x64", "128x64", "128x32"])
@pytest.mark.parametrize("case", _KS_CLASS_CASES, ids=lambda c: "%dx%dx%dx%d_k%d_r%d_s%d_p%d_d%d" % c)
def test_split_conv_every_tile_class(case, cls, F, dev):
    """conv_split.hip's tile classes forced through sscg_conv_desc.tuning: forward with the fused normalisation statistics, and the data
    gradient, against torch fp64."""
    n, c, h, w, k, r, s, p, d = case
    g = torch.Generator().manual_seed(sum(case) + cls)
    x = torch.randn(n, c, h, w, generator=g, dtype=torch.float64)
    wt = torch.randn(k, c, r, r, generator=g, dtype=torch.float64) * (1.0 / (c * r * r) ** 0.5)
    xr = x.clone().requires_grad_(True)
    yr = TF.conv2d(xr, wt, None, s, p, d)
    gy = torch.randn(yr.shape, generator=g, dtype=torch.float64)
    dxr = torch.autograd.grad(yr, xr, gy)[0]
    F.set_conv_precision("f32s")
    old = F.tuning(tile_class=cls)
    try:
        xg, wg = gpu(x, dev), gpu(wt, dev)
        L = n * yr.shape[2] * yr.shape[3]
        yg, cs = F.conv2d_fwd(xg, wg, None, s, p, d, stats=(1, L))
        if cs is not None:
            mean, rstd = F.norm_stats_from_conv(cs, (1, L, k), 1e-5)
        if k < 32:              # (a head's data gradient stays on the exact kernel family, whose tile classes this code does not name)
            F.TUNING[0] = 0
        dx = F.conv2d_dgrad(gpu(gy, dev), F.dgrad_operand(wg, x.shape, s, p, d), x.shape, wt.shape, s, p, d)
    finally:
        F.TUNING[0], F.WGRAD_TUNING[0] = old
        F.set_conv_precision("f32")
    tol = 2e-6 if (cls in (1, 2, 3) or k < 32) else 5e-6       # (wave tiles up to 32 x 64 carry two accumulator sets: conv_split.hip KS_ACC2)
    assert rel_err(yg, yr) < tol
    assert rel_err(dx, dxr) < tol
    if cs is not None:
        m64 = yr.detach().mean((0, 2, 3))
        v64 = yr.detach().var((0, 2, 3), unbiased=False)
        assert float(((mean[0].double().cpu() - m64) / v64.sqrt()).abs().max()) < 2e-6
        assert float((rstd[0].double().cpu() * (v64 + 1e-5).sqrt() - 1).abs().max()) < 2e-6


@pytest.mark.parametrize("case", [(2, 21, 64, 64, 64, 7, 2, 3, 0), (2, 21, 32, 32, 64, 7, 1, 3, 1), (3, 20, 17, 23, 64, 3, 1, 1, 0)],
                         ids=["deeplab_conv1_onehot", "resnet_stem_reflect", "ragged_20"])
def test_21_channel_stems_run_zero_padded_on_the_split_contraction(case, F, dev):
    """arch/generators.py:73,373 on a 21-channel one-hot / softmax map: in the split mode the convolution runs over 32 source channels
    (zero channels against zero weights: sscg_resize_channels) on the bf16 matrix cores instead of the exact kernel's ragged path;
    the data gradient's extra channels are cut again.  Forward (with the fused normalisation statistics), data gradient and the
    unchanged exact weight gradient against torch fp64; the padded weight copy follows an in-place update of the weight."""
    n, c, h, w, k, r, s, p, reflect = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(n, c, h, w, generator=g, dtype=torch.float64)
    wt = torch.randn(k, c, r, r, generator=g, dtype=torch.float64) * (1.0 / (c * r * r) ** 0.5)
    xr, wr = x.clone().requires_grad_(True), wt.clone().requires_grad_(True)
    yr = ref_conv(xr, wr, None, s, p, 1, reflect, 0)
    gy = torch.randn(yr.shape, generator=g, dtype=torch.float64)
    dxr, dwr = torch.autograd.grad(yr, (xr, wr), gy)
    calls = []
    real = F.resize_channels
    F.resize_channels = lambda t, cn: (calls.append((t.shape[1], cn)), real(t, cn))[1]
    F.set_conv_precision("f32s")
    try:
        xg, wg = gpu(x, dev), gpu(wt, dev)
        L = n * yr.shape[2] * yr.shape[3]
        yg, cs = F.conv2d_fwd(xg, wg, None, s, p, 1, F.PAD_REFLECT if reflect else F.PAD_ZEROS, stats=(1, L))
        assert (c, 32) in calls, calls                           # activation and weight were padded
        assert rel_err(yg, yr) < 2e-6
        if cs is not None:
            mean, rstd = F.norm_stats_from_conv(cs, (1, L, k), 1e-5)
            assert float(((mean[0].double().cpu() - yr.detach().mean((0, 2, 3))) / yr.detach().var((0, 2, 3), unbiased=False).sqrt()).abs().max()) < 2e-6
        if not reflect:
            dx = F.conv2d_dgrad_param(gpu(gy, dev), wg, x.shape, wt.shape, s, p, 1)
            assert (32, c) in calls, calls                       # the 32-channel gradient was cut back
            assert tuple(dx.shape) == tuple(x.shape) and rel_err(dx, dxr) < 2e-6
        dw = F.conv2d_wgrad(xg, gpu(gy, dev), wt.shape, s, p, 1, F.PAD_REFLECT if reflect else F.PAD_ZEROS)
        assert rel_err(dw, dwr) < 5e-6
        wg.mul_(2.0)                                             # torch rewrites the weight: the cached padded copy must follow
        y2 = F.conv2d_fwd(xg, wg, None, s, p, 1, F.PAD_REFLECT if reflect else F.PAD_ZEROS)
        assert rel_err(y2, 2.0 * yr) < 2e-6
    finally:
        F.resize_channels = real
        F.set_conv_precision("f32")


@pytest.mark.parametrize("with_addend", [False, True], ids=["plain", "fan_in_joined"])
def test_21_channel_head_data_gradient_runs_zero_padded_on_the_split_contraction(with_addend, F, dev):
    """The DeepLab classifier (arch/generators.py:373,388: 2048 -> 21, dilation 6 / 12; both classifiers read one tensor, so the
    second data gradient also joins the first's result): dy and the filters padded to 32 output channels, on the split contraction,
    against torch fp64."""
    n, c, h, w, k, r, p, d = 2, 2048, 9, 9, 21, 3, 6, 6
    g = torch.Generator().manual_seed(77 + with_addend)
    wt = torch.randn(k, c, r, r, generator=g, dtype=torch.float64) * (1.0 / (c * r * r) ** 0.5)
    xr = torch.randn(n, c, h, w, generator=g, dtype=torch.float64).requires_grad_(True)
    yr = TF.conv2d(xr, wt, None, 1, p, d)
    gy = torch.randn(yr.shape, generator=g, dtype=torch.float64)
    dxr = torch.autograd.grad(yr, xr, gy)[0]
    add = torch.randn(n, c, h, w, generator=g, dtype=torch.float64) if with_addend else None
    calls = []
    real = F.resize_channels
    F.resize_channels = lambda t, cn: (calls.append((t.shape[1], cn)), real(t, cn))[1]
    F.set_conv_precision("f32s")
    try:
        out = F.conv2d_dgrad_param(gpu(gy, dev), gpu(wt, dev), (n, c, h, w), wt.shape, 1, p, d, addend=gpu(add, dev) if with_addend else None)
    finally:
        F.resize_channels = real
        F.set_conv_precision("f32")
    assert calls == [(21, 32)], calls
    if with_addend:
        dx, rec, joined = out
        assert joined and rec is None
        assert rel_err(dx, dxr + add) < 2e-6
    else:
        assert rel_err(out, dxr) < 2e-6
