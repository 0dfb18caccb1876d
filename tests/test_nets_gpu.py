"""Network- and step-level parity of the HIP path against the golden vectors that tests/golden/gen_golden.py took
from the real reference (tests/golden/), plus the live oracle.  Tolerances follow SURVEY App. D:
InstanceNorm nets are held to 1e-4 against the fp64 golden; DeepLab-chained quantities are held to
k x (the reference's own fp32-vs-fp64 distance), never looser than north_star's 1e-3."""
import contextlib
import io
import json
import os

import numpy as np
import pytest
import torch

from conftest import load_sub
from oracle import fixtures as FX
from oracle import step as ostep

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


@pytest.fixture(scope="module")
def gold():
    meta = json.load(open(os.path.join(GOLD, "meta.json")))
    return meta, np.load(os.path.join(GOLD, "g2_nets.npz")), np.load(os.path.join(GOLD, "g3_step.npz"))


def build(kind, args, dev):
    arch = load_sub("arch")
    if kind in ("deeplab", "resnet_9blocks", "resnet_9blocks_softmax", "unet_128"):
        return quiet(arch.define_Gen, args[0], args[1], 64, kind, norm="instance", use_dropout=False, gpu_ids=[dev.index or 0])
    return quiet(arch.define_Dis, args[0], 64, kind, 3, norm="instance", gpu_ids=[dev.index or 0])


def rel_l2(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


# A ReLU whose input lies within fp32 rounding of zero flips its mask when the summation order of the producing conv
# changes (expected for about one of the ~1e6 activations of these nets; the reference's own fp32 run just happened to see
# none).  One flip moves the gradient of a handful of pixels by ~1e-2 of the tensor's maximum - the gradient is
# discontinuous there, so this is noise, not error - but it cannot move the tensor as a whole.  Gradients are therefore
# held to the strict max-norm bound, OR (flip metric, as for the post-step weights below) to a rel-L2 error within the
# strict bound with the max-norm excess confined to FLIP_BOUND.  One plan, no retries.
FLIP_BOUND = 5e-2


def _grad_ok(e_max, e_l2, tol):
    return e_max < tol or (e_l2 < tol and e_max < FLIP_BOUND)


@pytest.mark.parametrize("net", FX.NETS, ids=[n[0] for n in FX.NETS])
def test_network_forward_backward_vs_reference_golden(net, gold, dev):
    meta, g2, _ = gold
    F = load_sub("functional")
    name, kind, args, xshape = net
    m = build(kind, args, dev)
    m.load_state_dict(FX.net_weights(name, kind, args), strict=True)
    m.train()
    x = FX.net_input(name, xshape).to(dev).requires_grad_(True)
    y = m(x)
    y64, y32 = g2[name + "/y/f64"], g2[name + "/y/f32"]
    noise = rel(y32, y64)                      # the reference's own fp32 error on this net
    tol = max(4 * noise, 1e-4) if kind == "deeplab" else 1e-4
    assert tol <= 2e-3
    e = rel(y, y64)
    print("%s fwd: hip-vs-f64 %.2e  ref32-vs-f64 %.2e" % (name, e, noise))
    assert e < tol
    gy = FX.net_grad_out(name, y.shape).to(dev)
    y.backward(F.to_nhwc(gy))
    dx64 = g2[name + "/dx/f64"]
    noise_dx = rel(g2[name + "/dx/f32"], dx64)
    e, e2 = rel(x.grad, dx64), rel_l2(x.grad, dx64)
    print("%s dx: hip-vs-f64 max %.2e rel-L2 %.2e  ref32-vs-f64 max %.2e" % (name, e, e2, noise_dx))
    # ReLU-mask flips make the reference's own fp32 input-gradient differ from fp64 by percents on DeepLab
    # (SURVEY App. D): the bound is relative to that measured noise, not a fixed 1e-3
    dx_tol = max(4 * noise_dx, 1e-4)
    assert _grad_ok(e, e2, dx_tol), "dx: max %.2e, rel-L2 %.2e, tol %.2e" % (e, e2, dx_tol)
    gn, gn32 = meta["g2"][name + "/grad_norms/f64"], meta["g2"][name + "/grad_norms/f32"]
    worst = worst_noise = 0.0
    for k, p in m.named_parameters():
        if k in gn:
            assert p.grad is not None, k
            worst = max(worst, abs(float(p.grad.double().norm()) - gn[k]) / max(gn[k], 1e-30))
            worst_noise = max(worst_noise, abs(gn32[k] - gn[k]) / max(gn[k], 1e-30))
        else:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
    print("%s worst grad-norm rel err %.2e (reference fp32 noise %.2e)" % (name, worst, worst_noise))
    assert worst < max(4 * worst_noise, 1e-3 if e >= dx_tol else 1e-4)     # a norm is an L2 quantity: a mask flip barely moves it
    if kind == "deeplab":
        sd = m.state_dict()
        assert rel(sd["bn1.running_mean"], g2[name + "/bn1_running_mean/f64"]) < 1e-4
        assert rel(sd["layer4.2.bn3.running_var"], g2[name + "/l4_running_var/f64"]) < 1e-3
        assert int(sd["bn1.num_batches_tracked"]) == 1
        for k in ("conv1.weight", "layer3.10.conv2.weight", "layer5.conv2d_list.1.bias"):
            g = dict(m.named_parameters())[k].grad
            gflat = F.to_nchw(g).flatten()[:4096] if g.dim() == 4 else g.flatten()[:4096]
            ref64 = g2["%s/d_%s/f64" % (name, k)]
            nz = rel(g2["%s/d_%s/f32" % (name, k)], ref64)
            assert rel(gflat, ref64) < max(4 * nz, 1e-4), k


def _make_model(tag, dev, as_written=True):
    C, dataset, H, Wd, B, steps = FX.STEP_CONFIGS[tag]
    md = load_sub("model")
    args = FX.make_args(dataset=dataset, crop_height=H, crop_width=Wd, batch_size=B, gpu_ids=[dev.index or 0],
                        checkpoint_dir="/tmp/sscg_test_ckpt_none", as_written=as_written)
    m = quiet(md.semisuper_cycleGAN, args)
    sds = FX.semisup_state_dicts(C, torch.float32, tag)
    for k, sd in sds.items():
        getattr(m, k).load_state_dict(sd, strict=True)
    return m, (C, H, Wd, B, steps)


# criteria of SURVEY App. D.4: one DeepLab pass deep -> 1e-3 direct; two passes deep -> k x reference noise
DIRECT = ("lab_loss_CE", "lab_loss_MSE", "img_gen_loss", "gt_gen_loss", "img_dis_loss", "gt_dis_loss")
CHAINED = ("img_cycle_loss", "gt_cycle_loss", "cycle_img_dis_loss")


@pytest.mark.parametrize("tag", ["s64", "s128"])
def test_training_steps_vs_reference_golden(tag, gold, dev):
    meta, _, g3 = gold
    info = meta["g3"][tag]
    np.random.seed(0)
    m, (C, H, Wd, B, steps) = _make_model(tag, dev)
    for s in range(steps):
        l_img, l_gt, unl_img = FX.step_batch(tag, s, C, H, Wd, B)
        out = m.step(l_img.to(dev), l_gt.to(dev), unl_img.to(dev))
        got = {k: float(v) for k, v in out.items()}
        ref32, ref64 = info["reference_f32"][s], info["oracle_f64"][s]
        noises = {k: abs(ref32[k] - ref64[k]) / abs(ref64[k]) for k in ostep.LOSS_KEYS}
        # After the first Adam update (|dw| = lr for every weight, sign set by gradients that are partly
        # noise) the reference's own fp32 and fp64 trajectories are percents apart (meta.json g3: up to 9e-2
        # at step 2), so from step 1 on the yardstick is the largest fp32-vs-fp64 gap of that step.
        step_noise = max(noises.values())
        for k in ostep.LOSS_KEYS:
            noise = noises[k]
            e64 = abs(got[k] - ref64[k]) / abs(ref64[k])
            e32 = abs(got[k] - ref32[k]) / abs(ref32[k])
            print("step %d %-20s hip %.7f ref32 %.7f f64 %.7f | e64 %.1e noise %.1e" % (s, k, got[k], ref32[k], ref64[k], e64, noise))
            if s == 0 and k in DIRECT:
                assert e64 < 1e-3, (s, k)
            elif s == 0:
                # two chained DeepLab passes with an argmax one-hot in between: the END-TO-END value is a noise amplifier (the
                # reference's own fp32 run sits up to 3.7e-3 from fp64 on it), bounded statistically against fp64
                # (oracle.fixtures.chained_loss_bound); the same three losses are held to 1e-3 TEACHER-FORCED in
                # tests/test_teacher_forced_gpu.py, where the first pass's noise cannot compound.
                assert e64 < FX.chained_loss_bound(k, noise), (s, k)
            else:
                assert e64 < max(4 * step_noise, 1e-3), (s, k)
    # post-step state against the fp64 trajectory
    for net in ("Gis", "Gsi", "Di", "Ds"):
        sd = getattr(m, net).state_dict()
        for k in ("conv1.weight", "layer3.5.conv2.weight", "bn1.running_mean", "layer4.2.bn3.running_var",
                  "dis_model.2.weight", "dis_model.5.bias"):
            key = "%s/%s/%s/f64" % (tag, net, k)
            if key in g3.files:
                F = load_sub("functional")
                t = sd[k]
                flat = (F.to_nchw(t) if t.dim() == 4 else t).flatten()[:2048]
                r64 = torch.from_numpy(g3[key]).double()
                r32 = torch.from_numpy(g3["%s/%s/%s/f32" % (tag, net, k)]).double()
                if "running" in k:
                    e, nz = rel(flat, r64), rel(r32, r64)
                    print("%s %s.%s: e64 %.1e noise %.1e" % (tag, net, k, e, nz))
                    assert e < max(4 * nz, 1e-3), (net, k)
                else:
                    # Adam moves every weight by ~lr per step; a gradient whose sign is decided by rounding noise
                    # flips the direction (2*lr apart).  Count such flips instead of bounding the max error.
                    d_hip = (flat.double().cpu() - r64).abs()
                    d_ref = (r32 - r64).abs()
                    flips_hip, flips_ref = float((d_hip > 1e-4).float().mean()), float((d_ref > 1e-4).float().mean())
                    print("%s %s.%s: flip fraction hip %.2e ref32 %.2e, non-flip max err %.1e" % (
                        tag, net, k, flips_hip, flips_ref, float(d_hip[d_hip <= 1e-4].max())))
                    assert flips_hip <= 4 * flips_ref + 5e-3, (net, k)
                    assert float(d_hip.max()) < 2.5 * steps * 2e-4, (net, k)
        if net in ("Gis", "Gsi"):
            assert int(sd["bn1.num_batches_tracked"]) == info_nbt(meta, tag, net)


def test_supervised_steps_vs_reference_golden(gold, dev):
    meta, _, _ = gold
    cfg = meta["g4"]["config"]
    md = load_sub("model")
    args = FX.make_args(dataset="acdc", crop_height=cfg["H"], crop_width=cfg["H"], batch_size=cfg["B"], gpu_ids=[dev.index or 0],
                        model="supervised_model", checkpoint_dir="/tmp/sscg_test_ckpt_none2")
    m = quiet(md.supervised_model, args)
    m.Gsi.load_state_dict(FX.supervised_state_dict(cfg["C"], torch.float32), strict=True)
    for s in range(cfg["steps"]):
        smp = [FX.synth_sample("sup/lab", s * cfg["B"] + b, cfg["C"], cfg["H"], cfg["H"]) for b in range(cfg["B"])]
        loss = float(m.step(torch.stack([a for a, _ in smp]).to(dev), torch.stack([g for _, g in smp]).to(dev)))
        r32, r64 = meta["g4"]["reference_f32"][s], meta["g4"]["oracle_f64"][s]
        print("supervised step %d: hip %.7f ref32 %.7f f64 %.7f" % (s, loss, r32, r64))
        assert abs(loss - r64) / r64 < max(4 * abs(r32 - r64) / r64, 1e-3)   # step 1 follows an Adam update (see above)


def info_nbt(meta, tag, net):
    return meta["g3"]["%s/%s/num_batches_tracked" % (tag, net)]


def test_grouped_batchnorm_layer_equals_two_calls(dev):
    """BatchNorm2d under arch.batch_groups(2) on two stacked batches == two calls of the layer: outputs, input gradients,
    affine gradients and the twice-advanced running statistics (fp64 statistics: equal to fp32 rounding)."""
    arch, ops = load_sub("arch"), load_sub("arch.ops")
    g = torch.Generator().manual_seed(2)
    mk = lambda: ops.BatchNorm2d(64).to(dev)
    a, b = mk(), mk()
    with torch.no_grad():
        a.weight.copy_(torch.rand(64, generator=g) + 0.5)
        a.bias.copy_(torch.randn(64, generator=g))
    b.load_state_dict(a.state_dict())
    x = (torch.randn(6, 64, 17, 19, generator=g) * 2 + 1).to(dev)
    gy = torch.randn(6, 64, 17, 19, generator=g).to(dev)
    xa = x.clone().requires_grad_(True)
    ya = torch.cat([a(xa[:3], 1), a(xa[3:], 1)], 0)          # act = ReLU
    ya.backward(gy)
    xb = x.clone().requires_grad_(True)
    with arch.batch_groups(2):
        yb = b(xb, 1)
    yb.backward(gy)
    assert rel(yb, ya) < 1e-6 and rel(xb.grad, xa.grad) < 1e-5
    assert rel(b.weight.grad, a.weight.grad) < 1e-5 and rel(b.bias.grad, a.bias.grad) < 1e-5
    assert rel(b.running_mean, a.running_mean) < 1e-6 and rel(b.running_var, a.running_var) < 1e-6
    assert a.batches_tracked() == b.batches_tracked() == 2


def test_grouped_batchnorm_pass_equals_two_separate_passes(dev):
    """arch.batch_groups(2): one DeepLab pass over two stacked batches vs two passes.  Same arithmetic, different fp32
    summation orders (tile/split plans depend on the row count), amplified by 101 BatchNorm layers: the bound is the
    reference's own fp32-vs-fp64 distance on this net (3e-4 .. 4e-4 forward)."""
    arch = load_sub("arch")
    torch.manual_seed(0)
    mk = lambda: quiet(arch.define_Gen, 3, 5, 64, "deeplab", norm="instance", use_dropout=False, gpu_ids=[dev.index or 0])
    a, b = mk(), mk()
    b.load_state_dict(a.state_dict())
    g = torch.Generator().manual_seed(1)
    x1, x2 = (torch.randn(2, 3, 65, 65, generator=g).to(dev) for _ in range(2))
    with torch.no_grad():
        y1, y2 = a(x1), a(x2)
        with arch.batch_groups(2):
            yb = b(torch.cat([x1, x2], 0))
    assert rel(yb[:2], y1) < 2e-3 and rel(yb[2:], y2) < 2e-3
    sa, sb = a.state_dict(), b.state_dict()
    for k in sa:
        if "running" in k:
            assert rel(sb[k], sa[k]) < 2e-3, k
        if k.endswith("num_batches_tracked"):
            assert int(sa[k]) == int(sb[k]) == 2, k


def test_bf16_contraction_step_stays_within_bf16_noise_of_the_reference(gold, dev):
    """`--dtype bf16` (sscg_set_conv_precision(1)): the first G+D step of the golden 64x64 configuration.  Kernel-level
    parity of this mode is exact against bf16-rounded operands (test_kernels_gpu.py); end to end the nine losses must sit
    within bf16 rounding noise (2^-8 per operand, amplified by DeepLab) of the fp64 reference trajectory."""
    meta, _, _ = gold
    F = load_sub("functional")
    np.random.seed(0)
    try:
        F.set_conv_precision("bf16c")
        m, (C, H, Wd, B, steps) = _make_model("s64", dev)
        l_img, l_gt, unl_img = FX.step_batch("s64", 0, C, H, Wd, B)
        got = {k: float(v) for k, v in m.step(l_img.to(dev), l_gt.to(dev), unl_img.to(dev)).items()}
    finally:
        F.set_conv_precision("f32")
    ref64 = meta["g3"]["s64"]["oracle_f64"][0]
    worst = 0.0
    for k in ostep.LOSS_KEYS:
        e = abs(got[k] - ref64[k]) / abs(ref64[k])
        worst = max(worst, e)
        print("%-20s bf16 %.6f f64 %.6f  rel %.1e" % (k, got[k], ref64[k], e))
        assert np.isfinite(got[k]) and e < 0.1, k
    assert worst > 1e-6          # the mode was engaged: not the fp32 result
