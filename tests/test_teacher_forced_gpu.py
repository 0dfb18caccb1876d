"""The three CHAINED first-step losses, teacher-forced (/root/reference model.py:408-415,432,452,455 and :501-502,
527-528,534): `img_cycle_loss`, `gt_cycle_loss`, `cycle_img_dis_loss` sit two DeepLab passes deep - the fp32 noise of the first pass
is amplified by the second (101 BatchNorm layers at batch 2), so in the end-to-end step even the REFERENCE's own fp32 arithmetic
misses 1e-3 against fp64 on them (tests/golden/g7_first_steps.json: up to 3.7e-3) and the end-to-end tests can only bound them
statistically.  Here the second pass is fed the fp64 oracle's first-pass outputs (rounded to fp32): every quantity is then ONE pass
deep, the chaos cannot compound, and north_star's 1e-3 is asserted directly - in both fp32 arithmetics of the build (`f32s` = the
split contraction `--dtype f32` resolves to, `f32x` = exact fp32 MFMA), at 64x64 and at the bench geometry 256x256.

The oracle runs live on the box's host cores as the checker (fp64 = the truth, fp32 = the reference's own arithmetic, printed
beside the build's error).  d(loss)/d(second-pass input) is compared too; a DeepLab input gradient is discontinuous in the ReLU /
max-pool masks, and the reference's own fp32 arithmetic sits 3-4 % (rel-L2) from fp64 on it (measured below, every run), so its
bound is relative: no worse than 1.5 x the reference arithmetic's own distance (1e-3 when that is smaller), and the gradient's L2
norm - which a handful of mask flips cannot move - within 1e-2."""
import numpy as np
import pytest
import torch

from conftest import load_sub
from oracle import fixtures as FX
from oracle import step as ostep
from test_nets_gpu import quiet, rel_l2

pytestmark = pytest.mark.gpu
LOSSES = ("img_cycle_loss", "gt_cycle_loss", "cycle_img_dis_loss")
GRADS = ("d_fake_gt", "d_fake_img")


def oracle_second_pass(tag, C, H, B):
    """fp64 first pass -> fp32-rounded forcing inputs -> the oracle's second pass in fp64 (truth) and fp32 (the reference's arithmetic)."""
    l_img, l_gt, unl_img = FX.step_batch(tag, 0, C, H, H, B)
    # (torch's CPU convolutions on these small maps lose time beyond ~32 threads - bench.py's cpu_baseline measured 52 s per step on 128
    # threads against 10.5 s on 64: the checker runs on at most 32)
    threads = torch.get_num_threads()
    torch.set_num_threads(min(threads, 32))
    try:
        o64 = ostep.SemiSupOracle(C, FX.semisup_state_dicts(C, torch.float64, tag), crop=(H, H))
        fake_img, fake_gt, _ = o64.first_pass(l_img.double(), l_gt, unl_img.double(), want_lab=False)
        fake_img, fake_gt = fake_img.float(), fake_gt.float()
        r64 = o64.second_pass(fake_img.double(), fake_gt.double(), l_gt, unl_img.double())
        o32 = ostep.SemiSupOracle(C, FX.semisup_state_dicts(C, torch.float32, tag), crop=(H, H))
        r32 = o32.second_pass(fake_img, fake_gt, l_gt, unl_img)
    finally:
        torch.set_num_threads(threads)
    return (l_img, l_gt, unl_img, fake_img, fake_gt), r64, r32


def compare(got, r64, r32, label):
    """Relative errors of one arithmetic against the fp64 oracle; asserts the bounds of the module docstring."""
    out = {}
    for k in LOSSES:
        out[k] = abs(float(got[k]) - r64[k]) / abs(r64[k])
        ref = abs(r32[k] - r64[k]) / abs(r64[k])
        print("  %-6s %-20s %.6f  fp64 %.6f  rel err %.2e  (reference fp32 arithmetic %.2e)" % (label, k, float(got[k]), r64[k], out[k], ref))
    for k in GRADS:
        g, g64, g32 = got[k], r64[k], r32[k]
        e, ref = rel_l2(g, g64), rel_l2(g32, g64)
        n = abs(float(torch.as_tensor(g).double().norm()) - float(g64.norm())) / float(g64.norm())
        out[k], out[k + "/ref"], out[k + "/norm"] = e, ref, n
        print("  %-6s %-20s rel-L2 %.2e  (reference fp32 arithmetic %.2e)  |g| rel err %.2e" % (label, k, e, ref, n))
    for k in LOSSES:
        assert out[k] < 1e-3, (label, k, out[k])
    for k in GRADS:
        assert out[k] <= max(1e-3, 1.5 * out[k + "/ref"]), (label, k, out[k], out[k + "/ref"])
        assert out[k + "/norm"] < 1e-2, (label, k, out[k + "/norm"])
    return out


@pytest.mark.parametrize("tag,H", [("smoke", 64), ("tf256", 256)], ids=["s64", "s256"])
def test_second_pass_teacher_forced_meets_1e_3(tag, H, dev):
    F = load_sub("functional")
    md = load_sub("model")
    C, B = 21, 2                             # per-rank B >= 2 is the only regime the reference supports, at 256x256 too
    (l_img, l_gt, unl_img, fake_img, fake_gt), r64, r32 = oracle_second_pass(tag, C, H, B)
    args = FX.make_args(dataset="voc2012", crop_height=H, crop_width=H, batch_size=B, gpu_ids=[dev.index or 0],
                        checkpoint_dir="/tmp/sscg_test_ckpt_tf", as_written=True)
    old = F.get_conv_precision()
    print()
    try:
        F.set_conv_precision("f32x")
        m = quiet(md.semisuper_cycleGAN, args)
        for mode in ("f32s", "f32x"):
            F.set_conv_precision(mode)
            for k, sd in FX.semisup_state_dicts(C, torch.float32, tag).items():
                getattr(m, k).load_state_dict(sd, strict=True)
            got = m.second_pass(fake_img.to(dev), fake_gt.to(dev), l_gt.to(dev), unl_img.to(dev))
            torch.cuda.synchronize()
            compare(got, r64, r32, mode)
    finally:
        F.set_conv_precision("f32" if old in ("f32", "f32s") else old)
