"""TEST INFRASTRUCTURE (see oracle/__init__.py) - functional CPU restatement of the reference networks.

Every function takes a flat state dict (the reference's own state_dict keys, which are the checkpoint
ABI - SURVEY 8(b)) and an input tensor, and evaluates the network with stock torch CPU ops in the
dtype of the input (fp32 or fp64).  Each function cites the reference lines it restates.
"""
from collections import OrderedDict

import torch
import torch.nn.functional as TF

EPS = 1e-5
BN_MOMENTUM = 0.1


# --------------------------------------------------------------------------- specs (key -> (shape, kind))
def _conv(spec, key, cout, cin, k, bias):
    spec[key + ".weight"] = ((cout, cin, k, k), "conv")
    if bias:
        spec[key + ".bias"] = ((cout,), "bias")


def _bn(spec, key, c):
    spec[key + ".weight"] = ((c,), "bn_weight")
    spec[key + ".bias"] = ((c,), "bn_bias")
    spec[key + ".running_mean"] = ((c,), "running_mean")
    spec[key + ".running_var"] = ((c,), "running_var")
    spec[key + ".num_batches_tracked"] = ((), "nbt")


DEEPLAB_LAYERS = ((64, 3, 1, 1), (128, 4, 2, 1), (256, 23, 1, 2), (512, 3, 1, 4))  # planes, blocks, stride, dilation


def deeplab_spec(in_c, out_c):
    """State-dict layout of define_Gen(netG='deeplab') = ResNet(Bottleneck, [3,4,23,3]) (arch/generators.py:384-441,510-511)."""
    s = OrderedDict()
    _conv(s, "conv1", 64, in_c, 7, False)
    _bn(s, "bn1", 64)
    inplanes = 64
    for li, (planes, blocks, stride, dil) in enumerate(DEEPLAB_LAYERS, start=1):
        for b in range(blocks):
            p = "layer%d.%d" % (li, b)
            _conv(s, p + ".conv1", planes, inplanes, 1, False)
            _bn(s, p + ".bn1", planes)
            _conv(s, p + ".conv2", planes, planes, 3, False)
            _bn(s, p + ".bn2", planes)
            _conv(s, p + ".conv3", planes * 4, planes, 1, False)
            _bn(s, p + ".bn3", planes * 4)
            if b == 0:  # generators.py:410-416: every first block has a downsample branch
                _conv(s, p + ".downsample.0", planes * 4, inplanes, 1, False)
                _bn(s, p + ".downsample.1", planes * 4)
            inplanes = planes * 4
    for i in range(4):  # four classifier convs exist; only the first two are ever used (generators.py:378-382)
        _conv(s, "layer5.conv2d_list.%d" % i, out_c, 2048, 3, True)
    return s


def resnet_gen_spec(in_c, out_c, ngf=64, n_blocks=9, norm="instance"):
    """ResnetGenerator (arch/generators.py:65-95).  With norm='instance' every conv has a bias and norms have no state."""
    bias = norm == "instance"
    s = OrderedDict()

    def block(idx, cout, cin, k):
        _conv(s, "res_model.%d.0" % idx, cout, cin, k, bias)
        if norm == "batch":
            _bn(s, "res_model.%d.1" % idx, cout)

    block(1, ngf, in_c, 7)
    block(2, ngf * 2, ngf, 3)
    block(3, ngf * 4, ngf * 2, 3)
    for b in range(n_blocks):
        i = 4 + b
        _conv(s, "res_model.%d.res_block.1.0" % i, ngf * 4, ngf * 4, 3, bias)
        if norm == "batch":
            _bn(s, "res_model.%d.res_block.1.1" % i, ngf * 4)
        # without dropout the second conv sits at index 3, with dropout at 4 (arch/ops.py:62-70)
    return s


def resnet_gen_spec_full(in_c, out_c, ngf=64, n_blocks=9, norm="instance", use_dropout=True):
    bias = norm == "instance"
    s = OrderedDict()

    def cnr(idx, cout, cin, k, transposed=False):
        key = "res_model.%d.0" % idx
        shape = (cin, cout, k, k) if transposed else (cout, cin, k, k)
        s[key + ".weight"] = (shape, "conv")
        if bias:
            s[key + ".bias"] = ((cout,), "bias")
        if norm == "batch":
            _bn(s, "res_model.%d.1" % idx, cout)

    cnr(1, ngf, in_c, 7)
    cnr(2, ngf * 2, ngf, 3)
    cnr(3, ngf * 4, ngf * 2, 3)
    second = 4 if use_dropout else 3
    for b in range(n_blocks):
        i = 4 + b
        _conv(s, "res_model.%d.res_block.1.0" % i, ngf * 4, ngf * 4, 3, bias)
        if norm == "batch":
            _bn(s, "res_model.%d.res_block.1.1" % i, ngf * 4)
        _conv(s, "res_model.%d.res_block.%d" % (i, second), ngf * 4, ngf * 4, 3, bias)
        if norm == "batch":
            _bn(s, "res_model.%d.res_block.%d" % (i, second + 1), ngf * 4)
    t = 4 + n_blocks
    cnr(t, ngf * 2, ngf * 4, 3, transposed=True)
    cnr(t + 1, ngf, ngf * 2, 3, transposed=True)
    _conv(s, "res_model.%d" % (t + 3), out_c, ngf, 7, True)
    return s


def unet_spec(in_c, out_c, num_downs=7, ngf=64, norm="instance"):
    """UnetGenerator (arch/generators.py:7-63).  Keys follow the nesting of UnetSkipConnectionBlock.model: the outermost block is
    [downconv, submodule, ReLU, upconv]; inner blocks [LeakyReLU, downconv, norm, submodule, ReLU, upconv, norm]; the innermost
    [LeakyReLU, downconv, ReLU, upconv, norm].  ConvTranspose2d weights are [Cin, Cout, 4, 4]."""
    bias = norm == "instance"
    s = OrderedDict()
    # (outer_nc, inner_nc, input_nc) from the outermost block inwards
    chain = [(out_c, ngf, in_c), (ngf, ngf * 2, ngf), (ngf * 2, ngf * 4, ngf * 2), (ngf * 4, ngf * 8, ngf * 4)]
    chain += [(ngf * 8, ngf * 8, ngf * 8)] * (num_downs - 5) + [(ngf * 8, ngf * 8, ngf * 8)]
    pre = "unet_model."
    for depth, (outer, inner, inp) in enumerate(chain):
        outermost, innermost = depth == 0, depth == len(chain) - 1
        if outermost:
            _conv(s, pre + "model.0", inner, inp, 4, bias)
            up = (pre + "model.3", inner * 2, outer, True)
            nxt = pre + "model.1."
        elif innermost:
            _conv(s, pre + "model.1", inner, inp, 4, bias)
            up = (pre + "model.3", inner, outer, bias)
            nxt = None
        else:
            _conv(s, pre + "model.1", inner, inp, 4, bias)
            if norm == "batch":
                _bn(s, pre + "model.2", inner)
            up = (pre + "model.5", inner * 2, outer, bias)
            nxt = pre + "model.3."
        pending = (up, pre, outermost, innermost, outer)
        chain[depth] = pending
        pre = nxt
    # the up-convolutions (and their norms) come AFTER the submodule's keys: emit them from the innermost block outwards
    for (key, cin, cout, b), bpre, outermost, innermost, outer in reversed(chain):
        s[key + ".weight"] = ((cin, cout, 4, 4), "conv")
        if b:
            s[key + ".bias"] = ((cout,), "bias")
        if not outermost and norm == "batch":
            _bn(s, bpre + ("model.4" if innermost else "model.6"), outer)
    return s


def pixel_dis_spec(in_c, ndf=64, norm="instance"):
    """PixelDiscriminator (arch/discriminators.py:66-80)."""
    bias = norm == "instance"
    s = OrderedDict()
    _conv(s, "dis_model.0", ndf, in_c, 1, True)
    _conv(s, "dis_model.2", ndf * 2, ndf, 1, bias)
    if norm == "batch":
        _bn(s, "dis_model.3", ndf * 2)
    _conv(s, "dis_model.5", 1, ndf * 2, 1, bias)
    return s


def nlayer_dis_spec(in_c, ndf=64, n_layers=3, norm="instance"):
    """NLayerDiscriminator / 70x70 PatchGAN (arch/discriminators.py:42-63)."""
    bias = norm == "instance"
    s = OrderedDict()
    _conv(s, "dis_model.0", ndf, in_c, 4, True)
    mult = 1
    idx = 2
    for n in range(1, n_layers):
        prev, mult = mult, min(2 ** n, 8)
        _conv(s, "dis_model.%d.0" % idx, ndf * mult, ndf * prev, 4, bias)
        if norm == "batch":
            _bn(s, "dis_model.%d.1" % idx, ndf * mult)
        idx += 1
    prev, mult = mult, min(2 ** n_layers, 8)
    _conv(s, "dis_model.%d.0" % idx, ndf * mult, ndf * prev, 4, bias)
    if norm == "batch":
        _bn(s, "dis_model.%d.1" % idx, ndf * mult)
    idx += 1
    _conv(s, "dis_model.%d" % idx, 1, ndf * mult, 4, True)
    return s


# --------------------------------------------------------------------------- bf16 storage emulation (tests of the bf16 path)
class _RoundBoth(torch.autograd.Function):
    """An activation stored as bfloat16: the value is rounded on the way forward, its gradient on the way back."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


class _RoundFwd(torch.autograd.Function):
    """bf16 operand copy of an fp32 tensor (weights, network inputs): rounded forward, gradient untouched."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundBwd(torch.autograd.Function):
    """fp32 network output whose gradient enters a bf16 contraction: identity forward, gradient rounded."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


class Bf16Emulation:
    """Rounding points of the MI355X build's bf16 mode: `a` = activation written to HBM as bf16
    (conv outputs, normalise+activation outputs), `w` = conv weight operand, `i` = fp32 network input read by a bf16
    contraction, `o` = fp32 head output.  The arithmetic between the rounding points stays in the dtype of the tensors (fp64
    in the tests), so what is emulated is exactly the information bf16 storage discards."""
    a = staticmethod(_RoundBoth.apply)
    w = staticmethod(_RoundFwd.apply)
    i = staticmethod(_RoundFwd.apply)
    o = staticmethod(_RoundBwd.apply)


class _NoRounding:
    a = w = i = o = staticmethod(lambda t: t)


_NOQ = _NoRounding()


# --------------------------------------------------------------------------- building blocks
def _bn_apply(sd, key, x, train):
    """nn.BatchNorm2d: batch statistics + running-stat EMA in train mode (running tensors updated in place)."""
    rm, rv = sd[key + ".running_mean"], sd[key + ".running_var"]
    y = TF.batch_norm(x, rm, rv, sd[key + ".weight"], sd[key + ".bias"], train, BN_MOMENTUM, EPS)
    if train:
        sd[key + ".num_batches_tracked"] += 1
    return y


def _norm(sd, key, x, norm, train):
    if norm == "instance":  # arch/ops.py:11 (affine=False, track_running_stats=False)
        return TF.instance_norm(x, eps=EPS)
    return _bn_apply(sd, key, x, train)


def bottleneck(sd, p, x, stride, dil, train, q=_NOQ):
    """Bottleneck.forward (arch/generators.py:345-365).  `q`: rounding points of a bf16 run (Bf16Emulation), default none."""
    out = q.a(TF.conv2d(x, q.w(sd[p + ".conv1.weight"]), None, stride))
    out = q.a(torch.relu(_bn_apply(sd, p + ".bn1", out, train)))
    out = q.a(TF.conv2d(out, q.w(sd[p + ".conv2.weight"]), None, 1, dil, dil))
    out = q.a(torch.relu(_bn_apply(sd, p + ".bn2", out, train)))
    out = q.a(TF.conv2d(out, q.w(sd[p + ".conv3.weight"])))
    out = _bn_apply(sd, p + ".bn3", out, train)
    if (p + ".downsample.0.weight") in sd:
        res = q.a(TF.conv2d(x, q.w(sd[p + ".downsample.0.weight"]), None, stride))
        res = q.a(_bn_apply(sd, p + ".downsample.1", res, train))
    else:
        res = x
    return q.a(torch.relu(out + res))


def _deeplab_stem(sd, x, train, q):
    y = q.a(TF.conv2d(q.i(x), q.w(sd["conv1.weight"]), None, 2, 3))
    y = q.a(torch.relu(_bn_apply(sd, "bn1", y, train)))
    return TF.max_pool2d(y, 3, 2, 1, ceil_mode=True)


def _deeplab_head(sd, y, q):
    out = q.o(TF.conv2d(y, q.w(sd["layer5.conv2d_list.0.weight"]), sd["layer5.conv2d_list.0.bias"], 1, 6, 6))
    return out + q.o(TF.conv2d(y, q.w(sd["layer5.conv2d_list.1.weight"]), sd["layer5.conv2d_list.1.bias"], 1, 12, 12))


def deeplab(sd, x, train=True, taps=None, q=_NOQ):
    """ResNet.forward (arch/generators.py:430-441) incl. the two-of-four classifier quirk (:378-382).
    `taps` (optional dict) receives the stage outputs for teacher-forced per-stage checks."""
    y = _deeplab_stem(sd, x, train, q)
    if taps is not None:
        taps["stem"] = y
    for li, (planes, blocks, stride, dil) in enumerate(DEEPLAB_LAYERS, start=1):
        for b in range(blocks):
            y = bottleneck(sd, "layer%d.%d" % (li, b), y, stride if b == 0 else 1, dil, train, q)
        if taps is not None:
            taps["layer%d" % li] = y
    return _deeplab_head(sd, y, q)


def deeplab_stage(sd, name, x, train=True, q=_NOQ):
    """One stage of `deeplab` on a given stage input (teacher forcing)."""
    if name == "stem":
        return _deeplab_stem(sd, x, train, q)
    if name == "layer5":
        return _deeplab_head(sd, x, q)
    li = int(name[-1])
    planes, blocks, stride, dil = DEEPLAB_LAYERS[li - 1]
    y = x
    for b in range(blocks):
        y = bottleneck(sd, "layer%d.%d" % (li, b), y, stride if b == 0 else 1, dil, train, q)
    return y


def resnet_generator(sd, x, n_blocks=9, tanh=True, norm="instance", use_dropout=True, train=True, dropout_masks=None, q=_NOQ):
    """ResnetGenerator.forward (arch/generators.py:73-95) with ResidualBlock (arch/ops.py:59-74).

    Dropout(0.5) is active whenever use_dropout (the frozen generators are never put in eval mode,
    SURVEY App. A).  `dropout_masks` is a list of keep-masks (already scaled by 2) - one per block;
    None with use_dropout=True draws from torch's RNG like the reference."""
    def cnr(idx, y, stride, pad):
        y = q.a(TF.conv2d(y, q.w(sd["res_model.%d.0.weight" % idx]), sd.get("res_model.%d.0.bias" % idx), stride, pad))
        return q.a(torch.relu(_norm(sd, "res_model.%d.1" % idx, y, norm, train)))

    y = TF.pad(q.i(x), (3, 3, 3, 3), mode="reflect")
    y = cnr(1, y, 1, 0)
    y = cnr(2, y, 2, 1)
    y = cnr(3, y, 2, 1)
    second = 4 if use_dropout else 3
    for b in range(n_blocks):
        i = 4 + b
        pre = "res_model.%d.res_block." % i
        h = TF.pad(y, (1, 1, 1, 1), mode="reflect")
        h = q.a(TF.conv2d(h, q.w(sd[pre + "1.0.weight"]), sd.get(pre + "1.0.bias")))
        h = q.a(torch.relu(_norm(sd, pre + "1.1", h, norm, train)))
        if use_dropout:
            if dropout_masks is not None:
                h = h * dropout_masks[b]
            else:
                h = TF.dropout(h, 0.5, True)
        h = TF.pad(h, (1, 1, 1, 1), mode="reflect")
        h = q.a(TF.conv2d(h, q.w(sd[pre + "%d.weight" % second]), sd.get(pre + "%d.bias" % second)))
        h = _norm(sd, pre + "%d" % (second + 1), h, norm, train)
        y = q.a(y + h)
    t = 4 + n_blocks
    for idx in (t, t + 1):
        y = q.a(TF.conv_transpose2d(y, q.w(sd["res_model.%d.0.weight" % idx]), sd.get("res_model.%d.0.bias" % idx), 2, 1, 1))
        y = q.a(torch.relu(_norm(sd, "res_model.%d.1" % idx, y, norm, train)))
    y = TF.pad(y, (3, 3, 3, 3), mode="reflect")
    y = q.o(TF.conv2d(y, q.w(sd["res_model.%d.weight" % (t + 3)]), sd["res_model.%d.bias" % (t + 3)]))
    return torch.tanh(y) if tanh else y


def unet_generator(sd, x, num_downs=7, norm="instance", train=True):
    """UnetGenerator.forward (arch/generators.py:7-63).  nn.LeakyReLU(0.2, True) at the head of every inner block's `down` works
    in place on the block's input, which is also the first operand of the skip concatenation (:44): the skip carries the
    ACTIVATED tensor.  (nn.ReLU(True) at the head of `up` rewrites the submodule's output, which nobody else reads.)"""
    def block(pre, h, depth):
        outermost, innermost = depth == 0, depth == num_downs - 1
        if outermost:
            y = TF.conv2d(h, sd[pre + "model.0.weight"], sd.get(pre + "model.0.bias"), 2, 1)
            y = block(pre + "model.1.", y, depth + 1)
            return TF.conv_transpose2d(torch.relu(y), sd[pre + "model.3.weight"], sd[pre + "model.3.bias"], 2, 1)
        ha = TF.leaky_relu(h, 0.2)
        y = TF.conv2d(ha, sd[pre + "model.1.weight"], sd.get(pre + "model.1.bias"), 2, 1)
        if innermost:
            y = TF.conv_transpose2d(torch.relu(y), sd[pre + "model.3.weight"], sd.get(pre + "model.3.bias"), 2, 1)
            y = _norm(sd, pre + "model.4", y, norm, train)
        else:
            y = _norm(sd, pre + "model.2", y, norm, train)
            y = block(pre + "model.3.", y, depth + 1)
            y = TF.conv_transpose2d(torch.relu(y), sd[pre + "model.5.weight"], sd.get(pre + "model.5.bias"), 2, 1)
            y = _norm(sd, pre + "model.6", y, norm, train)
        return torch.cat([ha, y], 1)

    return block("unet_model.", x, 0)


def pixel_discriminator(sd, x, norm="instance", train=True, q=_NOQ):
    """PixelDiscriminator.forward (arch/discriminators.py:66-80)."""
    y = q.a(TF.leaky_relu(TF.conv2d(q.i(x), q.w(sd["dis_model.0.weight"]), sd["dis_model.0.bias"]), 0.2))
    y = q.a(TF.conv2d(y, q.w(sd["dis_model.2.weight"]), sd.get("dis_model.2.bias")))
    y = q.a(TF.leaky_relu(_norm(sd, "dis_model.3", y, norm, train), 0.2))
    return q.o(TF.conv2d(y, q.w(sd["dis_model.5.weight"]), sd.get("dis_model.5.bias")))


def nlayer_discriminator(sd, x, n_layers=3, norm="instance", train=True, q=_NOQ):
    """NLayerDiscriminator.forward (arch/discriminators.py:42-63)."""
    y = q.a(TF.leaky_relu(TF.conv2d(q.i(x), q.w(sd["dis_model.0.weight"]), sd["dis_model.0.bias"], 2, 1), 0.2))
    idx = 2
    for n in range(1, n_layers + 1):
        stride = 2 if n < n_layers else 1
        y = q.a(TF.conv2d(y, q.w(sd["dis_model.%d.0.weight" % idx]), sd.get("dis_model.%d.0.bias" % idx), stride, 1))
        y = q.a(TF.leaky_relu(_norm(sd, "dis_model.%d.1" % idx, y, norm, train), 0.2))
        idx += 1
    return q.o(TF.conv2d(y, q.w(sd["dis_model.%d.weight" % idx]), sd["dis_model.%d.bias" % idx], 1, 1))


# block-level restatements of arch/ops.py:40-57 (used by the block goldens)
def conv_norm_act(w, b, x, stride, pad, norm, act, slope=0.2, transposed=False, out_pad=0):
    if transposed:
        y = TF.conv_transpose2d(x, w, b, stride, pad, out_pad)
    else:
        y = TF.conv2d(x, w, b, stride, pad)
    y = TF.instance_norm(y, eps=EPS) if norm == "instance" else y
    return torch.relu(y) if act == "relu" else TF.leaky_relu(y, slope)


# utils.Vgg16 / utils.perceptual_loss (utils.py:145-208).  Pinned (tests/golden/gen_golden.py g6_perceptual): the reference's own
# function run on the CPU - torchvision.models.vgg16 stubbed by torchvision's published configuration-D layer table built from torch.nn
# with keyed weights (the pretrained ImageNet weights are not in the image), Module.cuda the identity - agrees with this restatement
# to 0.0 relative (loss and gradient, fp32 and fp64); golden: g6_perceptual.npz.
VGG16_CONVS = ((0, 3, 64), (2, 64, 64), (5, 64, 128), (7, 128, 128), (10, 128, 256), (12, 256, 256), (14, 256, 256),
               (17, 256, 512), (19, 512, 512), (21, 512, 512))
_VGG_SLICE = {0: 1, 2: 1, 5: 2, 7: 2, 10: 3, 12: 3, 14: 3, 17: 4, 19: 4, 21: 4}


def vgg16_relu2_2(sd, x):
    """slice1 = conv-relu-conv-relu, slice2 = maxpool(2,2)-conv-relu-conv-relu (features[0:9])."""
    def cr(h, idx):
        k = "slice%d.%d" % (_VGG_SLICE[idx], idx)
        return torch.relu(TF.conv2d(h, sd[k + ".weight"], sd[k + ".bias"], 1, 1))
    h = cr(cr(x, 0), 2)
    h = TF.max_pool2d(h, 2, 2)
    return cr(cr(h, 5), 7)


def perceptual_loss(sd, x, y):
    """utils.py:181-208 as written (u * std + mean after x / 2 + 1 / 2)."""
    mean = torch.tensor([0.485, 0.456, 0.406], dtype=x.dtype).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225], dtype=x.dtype).view(1, 3, 1, 1)
    u = (x * 0.5 + 0.5) * std + mean
    v = (y * 0.5 + 0.5) * std + mean
    return TF.mse_loss(vgg16_relu2_2(sd, v), vgg16_relu2_2(sd, u))
