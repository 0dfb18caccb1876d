"""TEST INFRASTRUCTURE (see oracle/__init__.py) - the shared recipe for golden inputs: which networks, which
shapes, which keyed streams.  tests/golden/gen_golden.py (build container, with the reference) and tests/ (anywhere)
regenerate identical inputs/weights from here, so fixtures only store outputs."""
import types

import torch

from . import nets
from . import weights as W

SEED = 20260928

# name, kind, ctor args, input shape
NETS = [
    ("deeplab_3_21", "deeplab", (3, 21), (2, 3, 64, 64)),
    ("deeplab_21_3", "deeplab", (21, 3), (2, 21, 64, 64)),
    ("resnet9_21_3", "resnet_9blocks", (21, 3), (2, 21, 32, 32)),
    ("resnet9sm_3_21", "resnet_9blocks_softmax", (3, 21), (2, 3, 32, 32)),
    ("pixel_3", "pixel", (3,), (2, 3, 32, 32)),
    ("pixel_21", "pixel", (21,), (2, 21, 32, 32)),
    ("nlayers_3", "n_layers", (3,), (2, 3, 64, 64)),
    ("unet128_3_2", "unet_128", (3, 2), (1, 3, 256, 256)),      # 7 downs: 256 -> 2x2 at the bottleneck (InstanceNorm needs > 1 pixel)
]

# Floor of the parity bound of the three CHAINED first-step losses (img_cycle_loss, gt_cycle_loss, cycle_img_dis_loss: two DeepLab
# passes with argmax / ReLU-mask flips in between).  The bound is k = 4 times the reference's own fp32-vs-fp64 distance on the
# loss, but that distance is ONE sample of a heavy-tailed noise (1e-4 .. 1.2e-3 on the goldens), so it needs a floor, and the floor
# must not sit below the noise itself: SURVEY App. D measured the reference moving by up to 2.4e-3 (img_cycle_loss) between
# OMP_NUM_THREADS=1 and 8, and every arithmetic variant of this build - exact fp32 MFMA or split contraction, any tile class -
# lands 0.2e-3 .. 2.4e-3 from the fp64 oracle on these three (profiles/r03_chained_loss_spread.txt).  Losses one pass deep keep 1e-3.
CHAINED_LOSS_FLOOR = 2.5e-3
CHAINED_LOSSES = ("img_cycle_loss", "gt_cycle_loss", "cycle_img_dis_loss")

STEP_CONFIGS = {  # tag -> (classes, dataset, H, W, batch, steps)
    "s64": (21, "voc2012", 64, 64, 2, 3),
    "s128": (21, "voc2012", 128, 128, 2, 1),
    "s256": (21, "voc2012", 256, 256, 2, 1),     # the bench geometry at the reference's default batch
}

# teacher-forced DeepLab stages: every stage is fed the reference's own (fp32-rounded) stage input
STAGE_NET = ("deeplab_3_21", "deeplab", (3, 21), (2, 3, 32, 32))
STAGES = ("stem", "layer1", "layer2", "layer3", "layer4", "layer5")

# per-epoch evaluation (model.py:555-574): Gsi.eval() -> interp -> softmax -> argmax -> runningScore
EVAL_CONFIG = dict(tag="ev", C=21, dataset="voc2012", H=64, W=64, batches=2, B=2)


def spec_for(kind, args):
    if kind == "deeplab":
        return nets.deeplab_spec(*args)
    if kind.startswith("resnet_9blocks"):
        return nets.resnet_gen_spec_full(args[0], args[1], 64, 9, "instance", use_dropout=False)
    if kind == "pixel":
        return nets.pixel_dis_spec(args[0])
    if kind == "unet_128":
        return nets.unet_spec(args[0], args[1], 7, 64, "instance")
    return nets.nlayer_dis_spec(args[0])


def oracle_forward(kind, sd, x, taps=None):
    if kind == "deeplab":
        return nets.deeplab(sd, x, True, taps)
    if kind == "resnet_9blocks":
        return nets.resnet_generator(sd, x, 9, True, "instance", False)
    if kind == "resnet_9blocks_softmax":
        return nets.resnet_generator(sd, x, 9, False, "instance", False)
    if kind == "pixel":
        return nets.pixel_discriminator(sd, x)
    if kind == "unet_128":
        return nets.unet_generator(sd, x, 7, "instance")
    return nets.nlayer_discriminator(sd, x)


def net_weights(name, kind, args, dtype=torch.float32):
    return W.fill_state_dict(spec_for(kind, args), SEED, dtype, prefix=name + "/")


def net_input(name, xshape, dtype=torch.float32):
    return W.uniform(SEED, name + "/x", xshape, -1.0, 1.0, dtype=dtype)


def net_grad_out(name, yshape, dtype=torch.float32):
    return W.normal(SEED, name + "/gy", tuple(yshape), dtype=dtype)


def synth_sample(stream, k, C, H, Wd, dtype=torch.float32):
    img = W.uniform(SEED, "%s/img/%d" % (stream, k), (3, H, Wd), -1.0, 1.0, dtype=dtype)
    gt = W.blob_labels(SEED, "%s/gt/%d" % (stream, k), 1, H, Wd, C, block=max(4, H // 8))[0]
    return img, gt


def step_batch(tag, s, C, H, Wd, B, dtype=torch.float32):
    """(l_img, l_gt, unl_img) of training step `s` for golden config `tag`."""
    lab = [synth_sample(tag + "/lab", s * B + b, C, H, Wd, dtype) for b in range(B)]
    unl = [synth_sample(tag + "/unl", s * B + b, C, H, Wd, dtype) for b in range(B)]
    return torch.stack([a for a, _ in lab]), torch.stack([g for _, g in lab]), torch.stack([a for a, _ in unl])


def make_args(**kw):
    """argparse.Namespace twin with the reference's defaults (main.py:12-44) for a parity run (no dropout)."""
    a = types.SimpleNamespace(epochs=2, decay_epoch=1, batch_size=2, lr=2e-4, gpu_ids=[], crop_height=64, crop_width=64,
                              lamda_img=0.5, lamda_gt=0.1, lamda_perceptual=0, lab_CE_weight=1, lab_MSE_weight=1,
                              lab_perceptual_weight=0, adversarial_weight=1.0, discriminator_weight=1.0, training=True,
                              testing=False, validation=False, model="semisupervised_cycleGAN", results_dir="/tmp/gg/res",
                              validation_dir="/tmp/gg/val", checkpoint_dir="/tmp/gg/ckpt", dataset="voc2012", norm="instance",
                              no_dropout=True, ngf=64, ndf=64, gen_net="deeplab", dis_net="fc_disc")
    a.__dict__.update(kw)
    return a


def semisup_state_dicts(C, dtype, tag):
    specs = {"Gis": nets.deeplab_spec(C, 3), "Gsi": nets.deeplab_spec(3, C), "Di": nets.pixel_dis_spec(3),
             "Ds": nets.pixel_dis_spec(C),
             "old_Gis": nets.resnet_gen_spec_full(C, 3, 64, 9, "instance", False),
             "old_Gsi": nets.resnet_gen_spec_full(3, C, 64, 9, "instance", False), "old_Di": nets.pixel_dis_spec(3)}
    # the generator works in float64 and casts last: the most recent (C, tag) set is kept in float64 (0.9 GB) - a test asks for the same
    # weights two or three times (HIP model, fp32 oracle, fp64 oracle) and the keyed fill costs ~9 s of host time per call
    if _SD_LAST.get("key") != (C, tag):
        _SD_LAST.clear()
        _SD_LAST["key"] = (C, tag)
        _SD_LAST["sds"] = {k: W.fill_state_dict(s, SEED, torch.float64, prefix="%s/%s/" % (tag, k)) for k, s in specs.items()}
    return {k: {kk: (v.clone() if v.dtype == dtype or not v.is_floating_point() else v.to(dtype)) for kk, v in sd.items()}
            for k, sd in _SD_LAST["sds"].items()}


_SD_LAST = {}


def supervised_state_dict(C, dtype):
    return W.fill_state_dict(nets.deeplab_spec(3, C), SEED, dtype, prefix="sup/Gsi/")


# ---------------------------------------------------------------------------------- block goldens (G1)
def _bn_keys(prefix, c):
    return {prefix + ".weight": (c,), prefix + ".bias": (c,), prefix + ".running_mean": (c,), prefix + ".running_var": (c,)}


BLOCK_SPECS = {
    "cnr": {"0.weight": (12, 8, 3, 3), "0.bias": (12,)},
    "cnl": {"0.weight": (12, 8, 4, 4), "0.bias": (12,)},
    "dcnr": {"0.weight": (8, 12, 3, 3), "0.bias": (12,)},
    "resblk": {"res_block.1.0.weight": (8, 8, 3, 3), "res_block.1.0.bias": (8,), "res_block.3.weight": (8, 8, 3, 3),
               "res_block.3.bias": (8,)},
    "bneck": dict([("conv1.weight", (4, 8, 1, 1))] + list(_bn_keys("bn1", 4).items()) + [("conv2.weight", (4, 4, 3, 3))] +
                  list(_bn_keys("bn2", 4).items()) + [("conv3.weight", (16, 4, 1, 1))] + list(_bn_keys("bn3", 16).items()) +
                  [("downsample.0.weight", (16, 8, 1, 1))] + list(_bn_keys("downsample.1", 16).items())),
    "cls": dict([("conv2d_list.%d.%s" % (i, k), s) for i in range(4) for k, s in (("weight", (5, 2048, 3, 3)), ("bias", (5,)))]),
}
BLOCK_INPUT = {"cls": ("g1/x2048", (2, 2048, 5, 5))}   # every other block uses ("g1/x8", (2, 8, 9, 10))


def block_state(name, dtype=torch.float32):
    """Keyed weights of a G1 block - the rule tests/golden/gen_golden.py applies to the reference module's state dict."""
    sd = {}
    for k, shape in BLOCK_SPECS[name].items():
        key = "g1/%s/%s" % (name, k)
        if len(shape) > 1:
            t = W.normal(SEED, key, shape, 0.0, 0.2)
        elif "running_var" in k or k.endswith("weight"):
            t = W.uniform(SEED, key, shape, 0.5, 1.5)
        else:
            t = W.normal(SEED, key, shape, 0.0, 0.1)
        if name == "cls":
            t = t * 0.1
        sd[k] = t.to(dtype)
    return sd


def block_input(name, dtype=torch.float32):
    key, shape = BLOCK_INPUT.get(name, ("g1/x8", (2, 8, 9, 10)))
    return W.normal(SEED, key, shape).to(dtype)


def block_grad_out(golden_name, yshape, dtype=torch.float32):
    return W.normal(SEED, "g1/%s/gy" % golden_name, tuple(yshape), dtype=dtype)


# ---- VGG16 perceptual loss (utils.py:145-208): keyed weights of torchvision's `features[0:23]` (He-scaled normal,
# small biases) in the build's Vgg16 state-dict layout (slice{1..4}.{features index}.{weight,bias}), and the two images of the golden
VGG_SHAPE = (2, 3, 24, 40)


def vgg_state_dict(dtype=torch.float32):
    from . import nets as _nets
    sd = {}
    for idx, cin, cout in _nets.VGG16_CONVS:
        k = "slice%d.%d" % (_nets._VGG_SLICE[idx], idx)
        sd[k + ".weight"] = W.normal(SEED, "vgg/" + k + ".weight", (cout, cin, 3, 3), 0.0, (2.0 / (cin * 9)) ** 0.5, dtype)
        sd[k + ".bias"] = W.normal(SEED, "vgg/" + k + ".bias", (cout,), 0.0, 0.05, dtype)
    return sd


def vgg_images(dtype=torch.float32):
    return (W.uniform(SEED, "vgg/x", VGG_SHAPE, -1.0, 1.0, dtype), W.uniform(SEED, "vgg/y", VGG_SHAPE, -1.0, 1.0, dtype))
