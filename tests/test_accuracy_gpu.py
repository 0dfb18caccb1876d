"""Is the split contraction (what `--dtype f32` computes with) "fp32-accurate" at NETWORK and STEP level, not only per kernel?

Measured against the fp64 oracle, side by side with the build's exact-fp32-MFMA mode (`f32x`) and with the reference's own fp32
arithmetic (the CPU oracle in fp32, pinned bit-exact against the reference): tests/golden/g7_first_steps.json holds the first-step
losses of six independent keyed weight sets (VOC 21 classes, 64x64, batch 2) in fp32 and fp64 (tests/golden/gen_first_steps.py).

The three CHAINED losses (two DeepLab passes with argmax / ReLU-mask flips in between) are a heavy-tailed noise: on these six
seeds the REFERENCE's fp32 arithmetic sits 0.06e-3 .. 3.7e-3 from fp64 (gt_cycle_loss: 3.7e-3, 3.3e-3; img_cycle_loss: 1.5e-3) -
north_star's 1e-3 cannot be asked of quantities the reference itself misses it on.  What can be asked, and is asserted here:
the split mode is no further from fp64 than the exact-fp32 mode, and neither is further than the reference's own arithmetic."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import load_sub
from oracle import fixtures as FX
from test_nets_gpu import CHAINED, DIRECT, build, quiet, rel_l2

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
SEEDS = ["ch%d" % i for i in range(6)]


def _first_step(m, tag, cfg, dev):
    """First-step losses of model `m` from the keyed weights `tag`.  One model serves every seed and both arithmetics: the nine
    losses of a step are taken before its updates, the state dicts carry the BatchNorm running statistics, and the image pools only
    hand back the current item while they fill."""
    for k, sd in FX.semisup_state_dicts(cfg["C"], torch.float32, tag).items():
        getattr(m, k).load_state_dict(sd, strict=True)
    l_img, l_gt, unl_img = FX.step_batch(tag, 0, cfg["C"], cfg["H"], cfg["W"], cfg["B"])
    np.random.seed(0)
    out = m.step(l_img.to(dev), l_gt.to(dev), unl_img.to(dev))
    m.sync_losses()
    out = {k: float(v) for k, v in out.items()}
    torch.cuda.synchronize()
    return out


def test_split_mode_is_as_close_to_fp64_as_exact_fp32_at_step_level(dev):
    """Measured (profiles/r05_accuracy.txt): pooled over 3 chained losses x 6 seeds, error / the reference's own worst error on that
    loss: exact fp32 MFMA rms 0.65 (max 1.42), split rms 0.27 (max 0.50; round 4, one accumulator chain per reduction: 0.72 / 1.93),
    the reference's fp32 arithmetic rms 0.58 (max 1.00); gt_cycle_loss reaches 5.3e-3 (exact), 1.4e-3 (split), 3.7e-3 (reference)."""
    F = load_sub("functional")
    md = load_sub("model")
    G = json.load(open(os.path.join(GOLD, "g7_first_steps.json")))
    cfg = G[SEEDS[0]]
    args = FX.make_args(dataset=cfg["dataset"], crop_height=cfg["H"], crop_width=cfg["W"], batch_size=cfg["B"], gpu_ids=[dev.index or 0],
                        checkpoint_dir="/tmp/sscg_test_ckpt_acc", as_written=True)
    err = {"f32x": {}, "f32s": {}, "ref": {}}
    old = F.get_conv_precision()
    try:
        F.set_conv_precision("f32x")
        m = quiet(md.semisuper_cycleGAN, args)
        for tag in SEEDS:
            r64, r32 = G[tag]["oracle_f64"], G[tag]["oracle_f32"]
            for mode in ("f32x", "f32s"):
                F.set_conv_precision(mode)
                got = _first_step(m, tag, G[tag], dev)
                for k in r64:
                    err[mode].setdefault(k, []).append(abs(got[k] - r64[k]) / abs(r64[k]))
            for k in r64:
                err["ref"].setdefault(k, []).append(abs(r32[k] - r64[k]) / abs(r64[k]))
    finally:
        F.set_conv_precision("f32" if old in ("f32", "f32s") else old)
    print()
    for k in CHAINED + DIRECT:
        print("%-20s " % k + "  ".join("%s med %.2e max %.2e" % (mo, float(np.median(err[mo][k])), max(err[mo][k])) for mo in ("ref", "f32x", "f32s")))
        print("%-20s " % "" + "  ".join("%s %s" % (mo, " ".join("%.1e" % e for e in err[mo][k])) for mo in ("f32x", "f32s")))
    # losses one DeepLab pass deep: north_star's 1e-3 against fp64, both modes, every seed (measured <= 4e-5)
    for k in DIRECT:
        assert max(err["f32x"][k]) < 1e-3 and max(err["f32s"][k]) < 1e-3, k
    # chained losses, pooled over the three losses and six seeds (18 draws per mode), each normalised by the noise scale of its loss
    # = the largest distance the REFERENCE's arithmetic shows on that loss over the six seeds
    scale = {k: max(err["ref"][k]) for k in CHAINED}
    pooled = {mo: np.array([e / scale[k] for k in CHAINED for e in err[mo][k]]) for mo in err}
    rms = {mo: float(np.sqrt(np.mean(pooled[mo] ** 2))) for mo in pooled}
    print("pooled chained error / reference noise scale: " + "  ".join("%s rms %.2f med %.2f max %.2f" % (mo, rms[mo], float(np.median(pooled[mo])), float(pooled[mo].max())) for mo in pooled))
    # the split mode is no further from fp64 than the exact mode (two draws of the same noise: rms within 1.5 x, single worst draw and
    # median within 2 x), and neither is further than the reference's own arithmetic by more than that
    assert rms["f32s"] <= 1.5 * rms["f32x"], rms
    assert float(np.median(pooled["f32s"])) <= 2.0 * float(np.median(pooled["f32x"])), pooled
    assert float(pooled["f32s"].max()) <= 2.0 * float(pooled["f32x"].max()), pooled
    assert rms["f32s"] <= 1.5 * rms["ref"] and rms["f32x"] <= 1.5 * rms["ref"], rms
    for k in CHAINED:       # per loss: the product's arithmetic (split) no further from fp64 than the reference's own worst draw on that
        # loss (measured round 5: at most 0.50 of it); the exact-fp32-MFMA mode (one rounding chain per reduction) within 2.5 x
        assert max(err["f32s"][k]) <= FX.chained_loss_bound(k), (k, err["f32s"][k], scale[k])
        assert max(err["f32x"][k]) <= 2.5 * scale[k], (k, err["f32x"][k], scale[k])
    assert rms["f32s"] <= rms["ref"], rms       # (0.27 against 0.58: two accumulator sets per wave tile, conv_split.hip KS_ACC2)
    # (no assertion that the build misses 1e-3 on these: a better summation order must not fail the suite.  The same three losses are
    # held to 1e-3 where that is attainable - teacher-forced, tests/test_teacher_forced_gpu.py.)
    print("worst chained draw: exact fp32 %.2e, split %.2e, the reference's fp32 %.2e" % tuple(max(max(err[mo][k]) for k in CHAINED) for mo in ("f32x", "f32s", "ref")))


@pytest.mark.parametrize("name", ["deeplab_3_21", "deeplab_21_3"])
def test_deeplab_forward_split_vs_exact_vs_fp64(name, dev):
    """DeepLab forward (101 BatchNorm layers at batch 2) against the fp64 golden, rel-L2 over the logits: the split mode's distance
    is within 1.25 x the exact mode's (+ 2e-7)."""
    F = load_sub("functional")
    g2 = np.load(os.path.join(GOLD, "g2_nets.npz"))
    net = [n for n in FX.NETS if n[0] == name][0]
    _, kind, args, xshape = net
    e = {}
    old = F.get_conv_precision()
    weights = FX.net_weights(name, kind, args)
    try:
        for mode in ("f32x", "f32s"):
            F.set_conv_precision(mode)
            m = build(kind, args, dev)
            m.load_state_dict(weights, strict=True)
            m.train()
            with torch.no_grad():
                y = m(FX.net_input(name, xshape).to(dev))
            e[mode] = rel_l2(y, g2[name + "/y/f64"])
    finally:
        F.set_conv_precision("f32" if old in ("f32", "f32s") else old)
    e["ref"] = rel_l2(g2[name + "/y/f32"], g2[name + "/y/f64"])
    print("%s forward rel-L2 vs fp64: reference fp32 %.2e, exact-fp32 MFMA %.2e, split %.2e" % (name, e["ref"], e["f32x"], e["f32s"]))
    assert e["f32s"] <= 1.25 * e["f32x"] + 2e-7
    assert e["f32s"] <= 1.1 * e["ref"] + 2e-7          # (measured round 5: 2.1e-4 / 2.1e-4 split against 2.7e-4 / 4.4e-4 for the reference's fp32 on the CPU;
                                                       #  round 4, before the two accumulator sets and the padded 21-channel stem: 3.2e-4 / 6.0e-4)
