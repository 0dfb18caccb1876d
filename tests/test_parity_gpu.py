"""Parity checks SURVEY App. D prescribes beyond whole-network comparisons (VERDICT r1 "tighten parity to App. D"):
  * block goldens of the reference's fusion units (tests/golden/g1_blocks.npz: y, dx, dW, db) on the HIP path;
  * DeepLab teacher-forced PER STAGE from the reference's own stage inputs (g2s_stages.npz) at a flat 1e-3;
  * the per-epoch evaluation of model.py:555-574 against label maps / mIoU the reference produced (g5_eval.npz);
  * a 256x256 training step (the bench geometry at the reference's default batch) against the reference's recorded losses;
  * a run long enough for the image pools to hand tensors of earlier steps to the overlapped discriminator stream."""
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
CL = torch.channels_last


def rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


@pytest.fixture(scope="module")
def meta():
    return json.load(open(os.path.join(GOLD, "meta.json")))


# ------------------------------------------------------------------------------------------ G1 blocks on the HIP path
def _block(name, dev):
    ops, gen = load_sub("arch.ops"), load_sub("arch.generators")
    IN = ops.get_norm_layer("instance")
    if name == "cnr":
        return ops.conv_norm_relu(8, 12, 3, 1, 1, norm_layer=IN, bias=True), "conv_norm_relu"
    if name == "cnl":
        return ops.conv_norm_lrelu(8, 12, 4, 2, 1, norm_layer=IN, bias=True), "conv_norm_lrelu"
    if name == "dcnr":
        return ops.dconv_norm_relu(8, 12, 3, 2, 1, 1, norm_layer=IN, bias=True), "dconv_norm_relu"
    if name == "resblk":
        return ops.ResidualBlock(8, IN, False, True), "residual_block"
    if name == "bneck":
        ds = ops.FusedSequential(ops.Conv2d(8, 16, 1, 1, bias=False), ops.BatchNorm2d(16))
        return gen.Bottleneck(8, 4, stride=1, dilation=2, downsample=ds), "bottleneck"
    return gen.Classifier_Module([6, 12, 18, 24], [6, 12, 18, 24], 5), "classifier"


@pytest.mark.parametrize("name", ["cnr", "cnl", "dcnr", "resblk", "bneck", "cls"])
def test_block_goldens_on_the_hip_path(name, dev):
    """The reference's fusion units (arch/ops.py:40-74, Bottleneck, Classifier_Module) with keyed weights: output, input
    gradient and every parameter gradient against the reference's fp64 run; tolerance 1e-4 (InstanceNorm blocks are benign,
    SURVEY App. D), 1e-3 where a BatchNorm over 180 samples amplifies fp32 rounding."""
    g1 = np.load(os.path.join(GOLD, "g1_blocks.npz"))
    F = load_sub("functional")
    m, gname = _block(name, dev)
    m = m.to(dev)
    sd = FX.block_state(name)
    if name == "resblk":       # the reference's ResidualBlock without dropout numbers its second conv res_block.3; ours keeps that key
        assert set(sd) == set(k for k in m.state_dict()), (sorted(sd), sorted(m.state_dict()))
    m.load_state_dict(sd, strict=False)
    m.train()
    for p in m.parameters():
        p.requires_grad_(True)
    x = FX.block_input(name).to(dev).requires_grad_(True)
    y = m(x)
    tol = 1e-3 if name == "bneck" else 1e-4
    assert rel(y, g1[gname + "/y/f64"]) < tol
    gy = FX.block_grad_out(gname, y.shape).to(dev)
    y.backward(F.to_nhwc(gy))
    assert rel(x.grad, g1[gname + "/dx/f64"]) < 10 * tol
    checked = 0
    wscale = max(float(np.abs(g1["%s/d_%s/f64" % (gname, k)]).max()) for k, p in m.named_parameters()
                 if "%s/d_%s/f64" % (gname, k) in g1.files and p.dim() == 4)
    for k, p in m.named_parameters():
        key = "%s/d_%s/f64" % (gname, k)
        if key in g1.files and p.grad is not None:
            g = F.to_nchw(p.grad) if p.grad.dim() == 4 else p.grad
            ref = g1[key]
            if float(np.abs(ref).max()) < 1e-9 * wscale:
                # the bias of a conv that feeds an InstanceNorm: its gradient is mathematically zero (the norm removes the
                # mean); the reference's fp64 value is 1e-15, ours fp32 rounding of the same cancellation
                assert float(g.abs().max()) < 1e-4 * wscale, k
            else:
                assert rel(g, ref) < 10 * tol, k
            checked += 1
    assert checked >= 1
    if name == "bneck":
        # the golden was taken after the generator ran the block twice (its fp32 and fp64 passes share the module)
        with torch.no_grad():
            m(x.detach())
        assert rel(m.bn2.running_mean, g1["bottleneck/running_mean_after/f32"]) < 1e-3


# ------------------------------------------------------------------------------------------ teacher-forced DeepLab stages
def test_deeplab_stages_teacher_forced(dev, meta):
    """each DeepLab stage (stem, layer1-4, classifier) is fed the REFERENCE's stage input (the fp32 rounding
    of its fp64 activations) and must reproduce the reference's fp64 stage output to 1e-3 - flat, no noise yardstick: a single
    stage has at most 23 Bottlenecks and the chaos of the 101-layer chain (App. D: 3-4e-4 end to end) cannot build up."""
    g = np.load(os.path.join(GOLD, "g2s_stages.npz"))
    arch = load_sub("arch")
    name, kind, args, xshape = FX.STAGE_NET
    m = quiet(arch.define_Gen, args[0], args[1], 64, kind, norm="instance", use_dropout=False, gpu_ids=[dev.index or 0])
    m.load_state_dict(FX.net_weights(name, kind, args), strict=True)
    m.train()
    fns = {"stem": m.stem, "layer1": m.layer1, "layer2": m.layer2, "layer3": m.layer3, "layer4": m.layer4, "layer5": m.layer5}
    with torch.no_grad():
        for st in FX.STAGES:
            x = torch.from_numpy(g[st + "/x"]).to(dev)
            y = fns[st](x)
            e = rel(y, g[st + "/y"])
            l2 = float((y.double().cpu() - torch.from_numpy(g[st + "/y"]).double()).norm() / torch.from_numpy(g[st + "/y"]).double().norm())
            print("stage %-7s in %s out %s: max rel err %.2e, rel-L2 %.2e" % (st, tuple(x.shape), tuple(y.shape), e, l2))
            assert tuple(y.shape) == tuple(meta["g2s"]["stages"][st])
            assert e < 1e-3, st


# ------------------------------------------------------------------------------------------ evaluation (N1/N2)
def test_evaluation_matches_the_references_label_maps(dev, meta):
    """model.py:555-574 run on the reference's modules by gen_golden.py: predicted label maps and mIoU.  Argmax is exact on
    identical logits; end to end a pixel may differ only where the reference's own top-2 margin is below the fp32 noise of
    the logits - every mismatching pixel is checked for that, and the mIoU must agree to 1e-3."""
    cfg = meta["g5_eval"]["config"]
    gold = np.load(os.path.join(GOLD, "g5_eval.npz"))
    md = load_sub("model")
    C, H, Wd = cfg["C"], cfg["H"], cfg["W"]
    args = FX.make_args(dataset=cfg["dataset"], crop_height=H, crop_width=Wd, batch_size=cfg["B"], gpu_ids=[dev.index or 0],
                        checkpoint_dir="/tmp/sscg_test_ckpt_ev5", as_written=True)
    m = quiet(md.semisuper_cycleGAN, args)
    m.Gsi.load_state_dict(FX.semisup_state_dicts(C, torch.float32, cfg["tag"])["Gsi"], strict=True)
    F = load_sub("functional")
    batches, preds = [], []
    for b in range(cfg["batches"]):
        smp = [FX.synth_sample(cfg["tag"] + "/val", b * cfg["B"] + i, C, H, Wd) for i in range(cfg["B"])]
        batches.append((torch.stack([a for a, _ in smp]), torch.stack([g for _, g in smp]), ["v"] * cfg["B"]))
    miou, class_iou = m.evaluate(batches)
    m.Gsi.eval()
    with torch.no_grad():
        for img, _, _ in batches:
            preds.append(F.argmax_index(F.softmax2d(m.interp(m.Gsi(img.to(dev))))).cpu().numpy())
    pred = np.stack(preds)
    mism = pred != gold["pred"]
    frac = float(mism.mean())
    print("evaluation: mIoU hip %.6f reference %.6f; label mismatch fraction %.2e" % (miou, meta["g5_eval"]["miou"], frac))
    assert frac < 2e-3
    if mism.any():
        assert float(gold["margin"][mism].max()) < 1e-3          # only near-ties of the reference's own softmax may flip
    assert abs(miou - meta["g5_eval"]["miou"]) < 1e-3
    for k, v in meta["g5_eval"]["class_iou"].items():
        if v is not None:
            assert abs(class_iou[int(k)] - v) < 5e-3, k


def test_validation_driver_writes_the_references_label_maps(dev, meta, tmp_path):
    """validation.py (reference validation.py:42-157) end to end on the synthetic val set of the evaluation golden: checkpoint file ->
    DeepLab Gsi in eval mode -> interp -> Softmax2d -> argmax -> paletted PNG.  The label maps read back from the PNGs must be the
    ones the REFERENCE's modules predicted (g5_eval.npz), pixel for pixel except where the reference's own top-2 softmax margin is
    below the fp32 noise of the logits; both drivers' paths (supervised: one PNG per image; semi-supervised: generated_labels)."""
    import sys
    from PIL import Image
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import validation as vdrv
    utils = load_sub("utils")
    cfg = meta["g5_eval"]["config"]
    gold = np.load(os.path.join(GOLD, "g5_eval.npz"))
    C, H, Wd, B = cfg["C"], cfg["H"], cfg["W"], cfg["B"]
    sds = FX.semisup_state_dicts(C, torch.float32, cfg["tag"])
    ck = tmp_path / "ckpt"
    os.makedirs(ck)
    utils.save_checkpoint({"epoch": 1, "Gsi": sds["Gsi"], "best_iou": 0.5}, str(ck / "latest_supervised_model.ckpt"))
    utils.save_checkpoint({"epoch": 1, "Gsi": sds["Gsi"], "Gis": sds["Gis"], "best_iou": 0.5}, str(ck / "latest_semisuper_cycleGAN.ckpt"))
    batches = []
    for b in range(cfg["batches"]):
        smp = [FX.synth_sample(cfg["tag"] + "/val", b * B + i, C, H, Wd) for i in range(B)]
        batches.append((torch.stack([a for a, _ in smp]), torch.stack([g for _, g in smp]), ["v%d_%d" % (b, i) for i in range(B)]))
    for model, sub in (("supervised_model", "supervised"), ("semisupervised_cycleGAN", os.path.join("unsupervised", "generated_labels"))):
        args = FX.make_args(dataset=cfg["dataset"], crop_height=H, crop_width=Wd, batch_size=B, gpu_ids=[dev.index or 0],
                            checkpoint_dir=str(ck), as_written=True)
        args.model, args.validation_dir = model, str(tmp_path / ("val_" + model))
        assert quiet(vdrv.validation, args, batches) == 0.5            # the checkpoint's best_iou comes back (validation.py:157)
        mism = total = 0
        for b in range(cfg["batches"]):
            for i in range(B):
                png = Image.open(os.path.join(args.validation_dir, sub, "v%d_%d.png" % (b, i)))
                assert png.mode == "P" and png.size == (Wd, H)
                got = np.asarray(png)
                bad = got != gold["pred"][b][i]
                mism += int(bad.sum())
                total += bad.size
                if bad.any():
                    assert float(gold["margin"][b][i][bad].max()) < 1e-3      # only near-ties of the reference's own softmax may flip
        print("%s: %d of %d pixels differ from the reference's label maps" % (model, mism, total))
        assert mism < 2e-3 * total
    sub = os.path.join(str(tmp_path / "val_semisupervised_cycleGAN"), "unsupervised")
    for d in ("regenerated_labels", "regenerated_image", "image_from_labels"):
        assert len(os.listdir(os.path.join(sub, d))) == cfg["batches"] * B


# ------------------------------------------------------------------------------------------ 256x256 step (bench geometry)
def test_training_step_256_vs_reference_golden(dev, meta):
    """one G+D step at 256x256 (batch 2) against the losses the reference recorded.  Criteria of App. D.4: the
    six losses one DeepLab pass deep within 1e-3; the three chained ones within 4x the reference's own fp32-vs-fp64 distance
    (floor: oracle.fixtures.CHAINED_LOSS_FLOOR, the reference's own run-to-run noise on these losses)."""
    info = meta["g3"]["s256"]
    C, dataset, H, Wd, B, steps = FX.STEP_CONFIGS["s256"]
    md = load_sub("model")
    args = FX.make_args(dataset=dataset, crop_height=H, crop_width=Wd, batch_size=B, gpu_ids=[dev.index or 0],
                        checkpoint_dir="/tmp/sscg_test_ckpt_256", as_written=True)
    m = quiet(md.semisuper_cycleGAN, args)
    for k, sd in FX.semisup_state_dicts(C, torch.float32, "s256").items():
        getattr(m, k).load_state_dict(sd, strict=True)
    np.random.seed(0)
    l_img, l_gt, unl_img = FX.step_batch("s256", 0, C, H, Wd, B)
    got = {k: float(v) for k, v in m.step(l_img.to(dev), l_gt.to(dev), unl_img.to(dev)).items()}
    ref32, ref64 = info["reference_f32"][0], info["oracle_f64"][0]
    for k in ostep.LOSS_KEYS:
        noise = abs(ref32[k] - ref64[k]) / abs(ref64[k])
        e64, e32 = abs(got[k] - ref64[k]) / abs(ref64[k]), abs(got[k] - ref32[k]) / abs(ref32[k])
        print("%-20s hip %.7f ref32 %.7f f64 %.7f | e64 %.1e e32 %.1e noise %.1e" % (k, got[k], ref32[k], ref64[k], e64, e32, noise))
        chained = k in ("img_cycle_loss", "gt_cycle_loss", "cycle_img_dis_loss")
        assert e64 < (FX.chained_loss_bound(k, noise) if chained else 1e-3), k    # (chained: 1e-3 teacher-forced in tests/test_teacher_forced_gpu.py)


# ------------------------------------------------------------------------------------------ pools meet the overlapped D stream
def test_pool_swaps_under_the_overlapped_discriminator_step(dev):
    """utils.Sample_from_Pool holds 50 batches; from step 51 on it hands back tensors of EARLIER steps (allocated on the main
    stream) to the discriminator step that runs on its own stream (overlap_d).  64 steps: the swap branch fires (seeded numpy
    RNG), the run must equal the serial schedule bit for bit and stay finite."""
    md = load_sub("model")
    res = []
    for overlap in (True, False):
        args = FX.make_args(dataset="voc2012", crop_height=32, crop_width=32, batch_size=2, gpu_ids=[dev.index or 0],
                            checkpoint_dir="/tmp/sscg_test_ckpt_pool", as_written=True)
        args.overlap_d = overlap
        m = quiet(md.semisuper_cycleGAN, args)
        for k, sd in FX.semisup_state_dicts(21, torch.float32, "pool").items():
            getattr(m, k).load_state_dict(sd, strict=True)
        np.random.seed(0)
        swaps = 0
        hist = []
        for s in range(64):
            l_img, l_gt, unl_img = FX.step_batch("pool", s % 4, 21, 32, 32, 2)
            before = [id(t) for t in m.pools[1].items]
            out = m.step(l_img.to(dev), l_gt.to(dev), unl_img.to(dev))
            if s >= 50 and [id(t) for t in m.pools[1].items] != before:
                swaps += 1
            if s % 8 == 7 or s >= 56:
                m.sync_losses()
                hist.append({k: float(v) for k, v in out.items()})
        torch.cuda.synchronize()
        assert all(np.isfinite(v) for h in hist for v in h.values()), (overlap, [(i, k, v) for i, h in enumerate(hist) for k, v in h.items() if not np.isfinite(v)][:6])
        res.append((hist, swaps, m.d_optimizer.arena.detach().clone(), m.g_optimizer.arena.detach()[::4099].clone()))
    assert res[0][1] >= 3, "the pool's swap branch never fired"
    assert res[0][1] == res[1][1]
    assert res[0][0] == res[1][0]
    assert torch.equal(res[0][2], res[1][2]) and torch.equal(res[0][3], res[1][3])
