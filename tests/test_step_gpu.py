"""Step-level behaviour on the MI355X beyond loss parity: evaluation path, checkpoint interchange of the
flat-arena optimiser, reflection-pad adjoint, determinism of the two-stream schedule."""
import contextlib
import os
import sys
import io

import numpy as np
import pytest
import torch
import torch.nn.functional as TF

from conftest import load_sub
from oracle import fixtures as FX
from oracle import nets
from oracle import step as ostep

pytestmark = pytest.mark.gpu


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def make_model(dev, tag="ck", as_written=True, H=64, ckpt="/tmp/sscg_test_ckpt_x"):
    md = load_sub("model")
    args = FX.make_args(dataset="voc2012", crop_height=H, crop_width=H, batch_size=2, gpu_ids=[dev.index or 0],
                        checkpoint_dir=ckpt, as_written=as_written)
    m = quiet(md.semisuper_cycleGAN, args)
    for k, sd in FX.semisup_state_dicts(21, torch.float32, tag).items():
        getattr(m, k).load_state_dict(sd, strict=True)
    return m, args


def test_reflect_pad_adjoint(dev):
    F = load_sub("functional")
    ops = load_sub("arch.ops")
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 8, 9, 11, generator=g, dtype=torch.float64)
    xr = x.clone().requires_grad_(True)
    yr = TF.pad(xr, (3, 3, 3, 3), mode="reflect")
    gy = torch.randn(yr.shape, generator=g, dtype=torch.float64)
    yr.backward(gy)
    xg = x.float().to(dev).requires_grad_(True)
    yg = ops.ReflectPadFn.apply(xg, 3)
    yg.backward(gy.float().to(dev).contiguous(memory_format=torch.channels_last))
    assert torch.equal(yg.detach().cpu().contiguous(), yr.detach().float())
    assert float((xg.grad.double().cpu() - xr.grad).abs().max()) < 1e-5


def test_resnet_generator_trains_through_reflection_padding(dev):
    """norm='instance' ResnetGenerator with a gradient-carrying input (arch surface, not on the as-written step)."""
    arch = load_sub("arch")
    m = quiet(arch.define_Gen, 3, 3, 16, "resnet_6blocks", "instance", False, [dev.index or 0])
    spec = nets.resnet_gen_spec_full(3, 3, 16, 6, "instance", False)
    from oracle import weights as W
    sd = W.fill_state_dict(spec, 5, torch.float64, prefix="rg/")
    m.load_state_dict({k: v.float() for k, v in sd.items()}, strict=True)
    x = W.uniform(5, "rg/x", (2, 3, 32, 32), -1, 1, dtype=torch.float64)
    xr = x.clone().requires_grad_(True)
    for v in sd.values():
        v.requires_grad_(True)
    yr = nets.resnet_generator(sd, xr, 6, True, "instance", False)
    yr.square().mean().backward()
    xg = x.float().to(dev).requires_grad_(True)
    yg = m(xg)
    F = load_sub("functional")
    loss = F.mse_const(yg, 0.0)
    loss.backward()
    rel = lambda a, b: float((a.double().cpu() - b).abs().max() / b.abs().max())
    assert rel(yg.detach(), yr.detach()) < 1e-4
    assert rel(xg.grad, xr.grad) < 1e-3
    w = dict(m.named_parameters())["res_model.5.res_block.1.0.weight"]
    assert rel(F.to_nchw(w.grad), sd["res_model.5.res_block.1.0.weight"].grad) < 1e-3


def test_evaluate_miou_matches_oracle(dev):
    m, args = make_model(dev, "ev")
    C, H = 21, 64
    batches = []
    for b in range(2):
        smp = [FX.synth_sample("ev/val", b * 2 + i, C, H, H) for i in range(2)]
        batches.append((torch.stack([a for a, _ in smp]), torch.stack([g for _, g in smp]), ["v"] * 2))
    miou, _ = m.evaluate(batches)
    sd = FX.semisup_state_dicts(C, torch.float64, "ev")["Gsi"]
    conf = np.zeros((C, C))
    mismatch = total = 0
    for img, gt, _ in batches:
        out = TF.interpolate(nets.deeplab(sd, img.double(), train=False), size=(H, H), mode="bilinear", align_corners=True)
        pred = out.argmax(1)
        conf += ostep.confusion(gt.squeeze(1).numpy(), pred.numpy(), C)
    ref = ostep.running_score(conf, "voc2012")[2]
    assert abs(miou - ref) < 5e-3, (miou, ref)     # argmax on near-tied fp32 logits may flip isolated pixels


def test_checkpoint_roundtrip_and_stock_adam_format(dev, tmp_path):
    m, args = make_model(dev, "ck", ckpt=str(tmp_path / "a"))
    batch = [t.to(dev) for t in FX.step_batch("ck", 0, 21, 64, 64, 2)]
    np.random.seed(0)
    m.step(*batch)
    g_sd = m.g_optimizer.state_dict()
    # torch.optim.Adam's layout: state[i] = {step, exp_avg, exp_avg_sq}; only parameters that received a gradient
    assert len(g_sd["state"]) == 216                                   # 108 + 108 tensors
    st = next(iter(g_sd["state"].values()))
    assert set(st) == {"step", "exp_avg", "exp_avg_sq"}
    # a stock torch optimiser accepts it
    ref_opt = torch.optim.Adam([torch.nn.Parameter(torch.empty_like(p)) for p in m.g_optimizer.trainable], lr=2e-4, betas=(0.5, 0.999))
    ref_opt.load_state_dict({"state": {k: {kk: (vv.cpu() if torch.is_tensor(vv) else vv) for kk, vv in v.items()} for k, v in g_sd["state"].items()},
                             "param_groups": [{**g_sd["param_groups"][0], "params": list(range(len(m.g_optimizer.trainable)))}]})
    ck = {"Gis": m.Gis.state_dict(), "Gsi": m.Gsi.state_dict(), "Di": m.Di.state_dict(), "Ds": m.Ds.state_dict(),
          "g_optimizer": g_sd, "d_optimizer": m.d_optimizer.state_dict()}
    path = str(tmp_path / "x.ckpt")
    torch.save(ck, path)
    m2, _ = make_model(dev, "other", ckpt=str(tmp_path / "b"))
    ck2 = torch.load(path, map_location="cpu")
    for k in ("Gis", "Gsi", "Di", "Ds"):
        getattr(m2, k).load_state_dict(ck2[k])
    m2.g_optimizer.load_state_dict(ck2["g_optimizer"])
    m2.d_optimizer.load_state_dict(ck2["d_optimizer"])
    for k in ("old_Gis", "old_Gsi", "old_Di"):
        getattr(m2, k).load_state_dict(getattr(m, k).state_dict())
    b1 = [t.to(dev) for t in FX.step_batch("ck", 1, 21, 64, 64, 2)]
    np.random.seed(1)
    l1 = {k: float(v) for k, v in m.step(*b1).items()}
    np.random.seed(1)
    l2 = {k: float(v) for k, v in m2.step(*b1).items()}
    for k in l1:
        assert abs(l1[k] - l2[k]) <= 1e-6 * abs(l1[k]), (k, l1[k], l2[k])
    assert m.Gsi.bn1.batches_tracked() == 6 and int(m.Gsi.state_dict()["bn1.num_batches_tracked"]) == 6


def test_many_steps_stay_finite_and_do_not_fault(dev):
    """Regression for an out-of-bounds read of masked weight lanes (1-channel heads) that only faulted when the
    per-step transposed-weight copies happened to land at the end of an allocator segment: 40 steps re-create
    those copies 80 times.  Also checks that the losses stay finite and the discriminator terms move."""
    m, _ = make_model(dev, "many")
    m.overlap_d = True          # the schedule main.py trains with: D step on its own stream, overlapping the next G forwards
    batch = [t.to(dev) for t in FX.step_batch("many", 0, 21, 64, 64, 2)]
    np.random.seed(0)
    first = last = None
    for it in range(40):
        out = m.step(*batch)
        if it == 0:
            m.sync_losses()
            first = {k: float(v) for k, v in out.items()}
    m.sync_losses()
    last = {k: float(v) for k, v in out.items()}
    assert all(np.isfinite(v) for v in last.values())
    assert last["lab_loss_CE"] < first["lab_loss_CE"]          # training on a fixed batch reduces the supervised loss


@pytest.mark.parametrize("cfg", [("cityscapes", 20, 64, 128), ("acdc", 4, 64, 64)], ids=["cityscapes_20c_64x128", "acdc_4c_64x64"])
def test_first_step_other_datasets_vs_oracle_golden(cfg, dev):
    """BASELINE configs 3/5 (Cityscapes, 20 classes, non-square crop) and the ACDC geometry (4 classes): first G+D
    step against the CPU oracle's losses on the same keyed weights / inputs (tests/golden/g7_first_steps.json, written by
    tests/golden/gen_first_steps.py; rounds 1-3 ran the oracle live on the GPU box's host: 55 s per case).  Exercises the 20- and
    4-channel (vectorised, non-fast-path) conv loaders.  Tolerances: SURVEY App. D (1e-3 direct; chained losses 4x the reference's own fp32-vs-fp64 distance)."""
    import json
    dataset, C, H, Wd = cfg
    md = load_sub("model")
    args = FX.make_args(dataset=dataset, crop_height=H, crop_width=Wd, batch_size=2, gpu_ids=[dev.index or 0],
                        checkpoint_dir="/tmp/sscg_test_ckpt_ds", as_written=True)
    m = quiet(md.semisuper_cycleGAN, args)
    tag = "ds_" + dataset
    G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g7_first_steps.json")))[tag]
    assert (G["C"], G["H"], G["W"], G["B"]) == (C, H, Wd, 2)
    for k, sd in FX.semisup_state_dicts(C, torch.float32, tag).items():
        getattr(m, k).load_state_dict(sd, strict=True)
    l_img, l_gt, unl_img = FX.step_batch(tag, 0, C, H, Wd, 2)
    np.random.seed(0)
    got = {k: float(v) for k, v in m.step(l_img.to(dev), l_gt.to(dev), unl_img.to(dev)).items()}
    ref, r64 = G["oracle_f32"], G["oracle_f64"]
    for k in ostep.LOSS_KEYS:
        noise = abs(ref[k] - r64[k]) / abs(r64[k])
        e = abs(got[k] - r64[k]) / abs(r64[k])
        print("%-20s hip %.6f oracle32 %.6f oracle64 %.6f  e64 %.1e noise %.1e" % (k, got[k], ref[k], r64[k], e, noise))
        chained = k in ("img_cycle_loss", "gt_cycle_loss", "cycle_img_dis_loss")
        # chained = two DeepLab passes with discrete argmax/ReLU-mask flips in between: bounded statistically against fp64
        # (oracle.fixtures.chained_loss_bound); held to 1e-3 teacher-forced in tests/test_teacher_forced_gpu.py
        assert (e < FX.chained_loss_bound(k, noise)) if chained else (e < 1e-3), k


def test_opt_in_nets_and_loss_variants(dev):
    """--honour_nets builds what --gen_net/--dis_net name (ResNet-6 generators, PatchGAN discriminators) and
    --variants adds the L1 image-cycle, lab_gt discriminator and VGG16 perceptual terms the reference has commented out and its dead Gaussian-noise branch.  No reference
    behaviour to pin (the reference never executes these): the step must run, stay finite and move every network."""
    md = load_sub("model")
    args = FX.make_args(dataset="voc2012", crop_height=64, crop_width=64, batch_size=2, gpu_ids=[dev.index or 0],
                        checkpoint_dir="/tmp/sscg_test_ckpt_n4", as_written=True)
    args.honour_nets, args.gen_net, args.dis_net, args.variants = 1, "resnet_6blocks", "n_layers", "l1_cycle,lab_gt_dis,perceptual,gauss_noise"
    args.lamda_perceptual, args.lab_perceptual_weight = 1, 1
    args.no_dropout = True
    torch.manual_seed(3)
    m = quiet(md.semisuper_cycleGAN, args)
    before = {k: next(getattr(m, k).parameters()).detach().clone() for k in ("Gis", "Gsi", "Di", "Ds")}
    l_img, l_gt, unl_img = FX.step_batch("n4", 0, 21, 64, 64, 2)
    out = m.step(l_img.to(dev), l_gt.to(dev), unl_img.to(dev))
    assert set(ostep.LOSS_KEYS) | {"img_cycle_l1", "gt_label_gen_loss", "img_cycle_loss_perceptual", "lab_loss_perceptual"} == set(out)
    assert all(bool(torch.isfinite(v)) for v in out.values())
    for k, w0 in before.items():
        assert not torch.equal(next(getattr(m, k).parameters()).detach(), w0), k


def test_honoured_nets_step_vs_oracle_golden(dev):
    """--honour_nets 1 --gen_net resnet_9blocks --dis_net n_layers (the reference parses both flags, main.py:43-44, and model.py:215-222
    never reads them): ResNet-9 generators (arch/generators.py:404-418) and PatchGAN discriminators (arch/discriminators.py:42-63) as the
    TRAINED nets, no dropout, against the oracle's step with the nets swapped (oracle/step.py gen_net= / dis_net=; golden "hn" of
    g7_first_steps.json).  InstanceNorm nets carry no chaos (oracle fp32 vs fp64 <= 3.3e-7 on every loss): all nine first-step losses
    are held to 1e-3 - the chained ones included - and the L2 norms of the generators' and discriminators' gradients to 1e-3."""
    import json
    md = load_sub("model")
    G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g7_first_steps.json")))["hn"]
    C, H, Wd = G["C"], G["H"], G["W"]
    args = FX.make_args(dataset="voc2012", crop_height=H, crop_width=Wd, batch_size=2, gpu_ids=[dev.index or 0],
                        checkpoint_dir="/tmp/sscg_test_ckpt_hn", as_written=True)
    args.honour_nets, args.gen_net, args.dis_net, args.no_dropout = 1, G["gen_net"], G["dis_net"], True
    m = quiet(md.semisuper_cycleGAN, args)
    for k, sd in FX.semisup_state_dicts(C, torch.float32, "hn", G["gen_net"], G["dis_net"]).items():
        getattr(m, k).load_state_dict(sd, strict=True)
    l_img, l_gt, unl_img = FX.step_batch("hn", 0, C, H, Wd, 2)
    np.random.seed(0)
    out = m.step(l_img.to(dev), l_gt.to(dev), unl_img.to(dev))
    m.sync_losses()
    torch.cuda.synchronize()
    got = {k: float(v) for k, v in out.items()}
    r64 = G["oracle_f64"]
    assert set(got) == set(r64)
    print()
    for k in r64:
        e = abs(got[k] - r64[k]) / abs(r64[k])
        print("%-20s hip %.6f oracle64 %.6f  e64 %.1e" % (k, got[k], r64[k], e))
        assert e < 1e-3, (k, e)
    for name, opt in (("g", m.g_optimizer), ("d", m.d_optimizer)):
        n = float(opt.grad.double().norm())
        n64 = G["%s_grad_norm_f64" % name]
        print("%s gradient norm: hip %.6e oracle64 %.6e rel %.1e" % (name, n, n64, abs(n - n64) / n64))
        assert abs(n - n64) / n64 < 1e-3, name


def test_variant_step_vs_oracle_golden(dev):
    """--variants l1_cycle,lab_gt_dis with the reference's default networks: the two loss terms the reference has commented out
    (model.py:453; :439,:447) against the oracle's restatement of those lines (oracle/step.py `variants=`, golden "var" of
    tests/golden/g7_first_steps.json): all nine losses + the two extra terms of the first step, and the L2 norm of the generators'
    gradient - what the extra terms change besides their own value."""
    import json
    md = load_sub("model")
    GA = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g7_first_steps.json")))
    G = GA["var"]
    C, H, Wd = G["C"], G["H"], G["W"]
    args = FX.make_args(dataset="voc2012", crop_height=H, crop_width=Wd, batch_size=2, gpu_ids=[dev.index or 0],
                        checkpoint_dir="/tmp/sscg_test_ckpt_var", as_written=True)
    args.variants = G["variants"]
    m = quiet(md.semisuper_cycleGAN, args)
    for k, sd in FX.semisup_state_dicts(C, torch.float32, "var").items():
        getattr(m, k).load_state_dict(sd, strict=True)
    l_img, l_gt, unl_img = FX.step_batch("var", 0, C, H, Wd, 2)
    np.random.seed(0)
    got = {k: float(v) for k, v in m.step(l_img.to(dev), l_gt.to(dev), unl_img.to(dev)).items()}
    torch.cuda.synchronize()
    ref, r64 = G["oracle_f32"], G["oracle_f64"]
    assert set(got) == set(r64) and {"img_cycle_l1", "gt_label_gen_loss"} <= set(got)
    scale = {k: max(abs(GA["ch%d" % i]["oracle_f32"][k] - GA["ch%d" % i]["oracle_f64"][k]) / abs(GA["ch%d" % i]["oracle_f64"][k]) for i in range(6))
             for k in FX.CHAINED_LOSSES}
    for k in r64:
        noise = abs(ref[k] - r64[k]) / abs(r64[k])
        e = abs(got[k] - r64[k]) / abs(r64[k])
        print("%-20s hip %.6f oracle32 %.6f oracle64 %.6f  e64 %.1e noise %.1e" % (k, got[k], ref[k], r64[k], e, noise))
        chained = k in FX.CHAINED_LOSSES
        if chained:
            # a NEW seed for the chained losses: its bound comes from the six-seed evidence of tests/test_accuracy_gpu.py, not from a
            # fixed floor - no draw beyond 2.5 x the worst distance to fp64 the REFERENCE's arithmetic shows on this loss over those
            # seeds (gt_cycle_loss: 3.7e-3; the build's two arithmetics reach 5.3e-3 / 7.1e-3 there), or 4 x this seed's own noise
            assert e < FX.chained_loss_bound(k), (k, e, noise, scale[k])
            continue
        if k == "img_cycle_l1":
            # taken on recon_img = Gis(Gsi(unl_img)) itself: two DeepLab passes deep with nothing smoothing it - the noise class of
            # gt_cycle_loss: held to THAT loss's bound (oracle.fixtures.chained_loss_bound: the worst distance the reference's own
            # fp32 arithmetic shows to fp64 over six seeds, 3.7e-3).  Measured on the round-6 tree: 2.1e-3 (the oracle's fp32 on
            # this seed: 7.4e-4); round 4's 2.5 x scale = 9.2e-3 is gone with the accumulator fix of round 5.
            assert e < FX.chained_loss_bound("gt_cycle_loss"), (k, e, noise)
            continue
        assert e < 1e-3, k
    gn = float(m.g_optimizer.grad.double().norm())
    n64, n32 = G["g_grad_norm_f64"], G["g_grad_norm_f32"]
    noise = abs(n32 - n64) / n64
    print("generator gradient norm: hip %.6e oracle32 %.6e oracle64 %.6e  rel %.1e (oracle noise %.1e)" % (gn, n32, n64, abs(gn - n64) / n64, noise))
    assert abs(gn - n64) / n64 < max(4 * noise, 1e-3)
