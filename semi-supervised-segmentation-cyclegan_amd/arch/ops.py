"""Operator surface of the reference's arch/ops.py:7-80, backed by libsscg.so.

Same factory names, argument order and state_dict keys (`0.weight`, `0.bias`, `1.*` inside a
conv_norm_* block; `res_block.{1.0,3|4}.*` inside a ResidualBlock), so checkpoints interchange with the
reference.  What differs is the execution: a block is evaluated as fused HIP launches
(conv [+bias] -> one statistics pass -> normalise+activation[+residual]), ReflectionPad2d is folded into
the conv's tile loader, and every tensor is channels-last in memory.
"""
import functools

import os

import torch
from torch import nn

from .. import functional as F
from .._lib import ACT_LRELU, ACT_NONE, ACT_RELU, ACT_TANH, PAD_REFLECT, PAD_ZEROS, SscgError

CL = torch.channels_last


# ----------------------------------------------------------------------------- leaf modules
class Conv2d(nn.Module):
    """nn.Conv2d twin (square kernels, as everywhere in the reference); weight is channels-last."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, bias=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding, self.dilation = kernel_size, stride, padding, dilation
        w = torch.empty(out_channels, in_channels, kernel_size, kernel_size).contiguous(memory_format=CL)
        self.weight = nn.Parameter(w)
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        with torch.no_grad():
            self.weight.copy_(torch.empty(w.shape).normal_(0.0, 0.02))

    head = False     # a network's last conv: its output stays fp32 in bf16 mode (set by the network constructors)

    def _geometry(self, x, reflect):
        """(x, pad, pad_mode) with nn.ReflectionPad2d folded into the conv's loader where possible."""
        if reflect:
            if self.padding != 0:
                raise SscgError("reflection padding folds only into an unpadded conv")
            if torch.is_grad_enabled() and x.requires_grad:
                # the frozen generators are the reference's only reflect users (model.py:225-228); a training
                # caller gets a materialised pad whose adjoint is a separate kernel
                return ReflectPadFn.apply(x, reflect), 0, PAD_ZEROS
            return x, reflect, PAD_REFLECT
        return x, self.padding, PAD_ZEROS

    def forward(self, x, reflect=0, act=ACT_NONE, slope=0.0):
        x, pad, mode = self._geometry(x, reflect)
        return F.conv2d(x, self.weight, self.bias, self.stride, pad, self.dilation, mode, act, slope, out_f32=self.head)

    def forward_stats(self, x, spec, reflect=0):
        """Convolution + the batch statistics of the normalisation layer described by `spec` (norm.stat_spec()), produced
        by the conv's epilogue.  Returns (y, (mean, rstd) or None)."""
        x, pad, mode = self._geometry(x, reflect)
        y, mean, rstd = F.conv2d_norm_stats(x, self.weight, self.bias, self.stride, pad, self.dilation, mode, spec)
        return y, ((mean, rstd) if mean is not None else None)

    def extra_repr(self):
        return "%d, %d, k=%d, s=%d, p=%d, d=%d" % (self.in_channels, self.out_channels, self.kernel_size, self.stride,
                                                  self.padding, self.dilation)


class ReflectPadFn(torch.autograd.Function):
    """Materialised nn.ReflectionPad2d with its adjoint (scatter-add expressed through the upsampling-free path:
    gradient of a pad is a sum of mirrored slices; evaluated with the weight-free identity conv kernels)."""

    @staticmethod
    def forward(ctx, x, pad):
        x = F.to_nhwc(x)
        ctx.pad = pad
        ctx.shape = tuple(x.shape)
        return F.reflect_pad(x, pad)

    @staticmethod
    def backward(ctx, dy):
        return F.reflect_pad_bwd(F.to_nhwc(dy), ctx.pad), None


class ConvTranspose2d(nn.Module):
    """nn.ConvTranspose2d twin; weight logical [Cin, Cout, k, k], channels-last memory."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, output_padding=0, bias=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding, self.output_padding = kernel_size, stride, padding, output_padding
        w = torch.empty(in_channels, out_channels, kernel_size, kernel_size).contiguous(memory_format=CL)
        self.weight = nn.Parameter(w)
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        with torch.no_grad():
            self.weight.copy_(torch.empty(w.shape).normal_(0.0, 0.02))

    head = False     # a network's last layer: fp32 output in bf16 mode (UnetGenerator's outermost up-convolution)

    def forward(self, x, act=ACT_NONE, slope=0.0):
        return F.conv_transpose2d(x, self.weight, self.bias, self.stride, self.padding, self.output_padding, act, slope,
                                  out_f32=self.head)


class InstanceNorm2d(nn.Module):
    """nn.InstanceNorm2d(affine=False, track_running_stats=False) (arch/ops.py:11): no state."""

    def __init__(self, num_features, eps=1e-5):
        super().__init__()
        self.num_features, self.eps = num_features, eps

    def stat_spec(self):
        return (True, self.eps, None, None, 0.0)

    def forward(self, x, act=ACT_NONE, slope=0.0, residual=None, stats=None):
        return F.instance_norm_act(x, act, slope, residual, self.eps, stats=stats)


_BATCH_GROUPS = [1]


class batch_groups:
    """`with batch_groups(k): net(torch.cat([x1, ..., xk]))` - every BatchNorm2d inside treats the input as k batches
    stacked along N: per-group batch statistics, running statistics advanced k times in order.  The result is what k
    separate forwards of the network compute, but every convolution sees k times the rows (the DeepLab maps of one
    batch of 8 are only 8712 rows, ~2 tiles per CU) and the pass costs one set of launches."""

    def __init__(self, k):
        self.k = int(k)

    def __enter__(self):
        self.prev = _BATCH_GROUPS[0]
        _BATCH_GROUPS[0] = self.k

    def __exit__(self, *exc):
        _BATCH_GROUPS[0] = self.prev
        return False


class BatchNorm2d(nn.Module):
    """nn.BatchNorm2d twin (affine, running statistics).  `num_batches_tracked` is counted on the host and
    written to its buffer when the state dict is taken, so a forward launches no bookkeeping kernel."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True):
        super().__init__()
        self.num_features, self.eps, self.momentum = num_features, eps, momentum
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self._pending = 0
        self._register_state_dict_hook(BatchNorm2d._flush_hook)

    @staticmethod
    def _flush_hook(module, state_dict, prefix, local_metadata):
        if module._pending:
            module.num_batches_tracked += module._pending
            module._pending = 0
            state_dict[prefix + "num_batches_tracked"] = module.num_batches_tracked
        return state_dict

    def _load_from_state_dict(self, *a, **k):
        self._pending = 0
        return super()._load_from_state_dict(*a, **k)

    def batches_tracked(self):
        return int(self.num_batches_tracked) + self._pending

    def stat_spec(self):
        """What a conv epilogue needs to produce this layer's statistics; None in eval mode (running statistics are used)."""
        if not self.training:
            return None
        groups = _BATCH_GROUPS[0]
        return (False if groups == 1 else int(groups), self.eps, self.running_mean, self.running_var, self.momentum)

    def forward(self, x, act=ACT_NONE, slope=0.0, residual=None, stats=None):
        groups = _BATCH_GROUPS[0]
        if self.training:
            self._pending += groups
        return F.batch_norm_act(x, self.weight, self.bias, self.running_mean, self.running_var, self.training,
                                self.momentum, self.eps, act, slope, residual, groups=groups, stats=stats)


class _Act(nn.Module):
    code, slope = ACT_NONE, 0.0

    def forward(self, x):
        return F.ActFn.apply(x, self.code, self.slope)


class ReLU(_Act):
    code = ACT_RELU

    def __init__(self, inplace=False):
        super().__init__()


class LeakyReLU(_Act):
    code = ACT_LRELU

    def __init__(self, negative_slope=0.01, inplace=False):
        super().__init__()
        self.slope = negative_slope


class Tanh(_Act):
    code = ACT_TANH


_DROPOUT_INSTANCES = [0]


class Dropout(nn.Module):
    """nn.Dropout: counter-hash mask (seed advances per call; torch's Philox stream cannot be matched, so
    parity runs use --no_dropout, SURVEY section 7 'hard parts').  Every instance draws from its own stream: the seed
    mixes the run seed (torch.initial_seed(), i.e. torch.manual_seed) with the instance's construction index, so the
    nine residual blocks of a generator - and the two frozen generators - see independent masks, as nn.Dropout's do."""

    def __init__(self, p=0.5):
        super().__init__()
        self.p = p
        self._calls = 0
        _DROPOUT_INSTANCES[0] += 1
        self.seed = ((torch.initial_seed() * 0x9E3779B1 + _DROPOUT_INSTANCES[0] * 0x85EBCA6B) & 0x7FFFFFFFFFF) | 1

    def forward(self, x):
        if not self.training or self.p == 0.0:
            return x
        self._calls += 1
        return F.DropoutFn.apply(x, self.p, (self.seed << 20) + self._calls)


class ReflectionPad2d(nn.Module):
    def __init__(self, padding):
        super().__init__()
        self.padding = padding

    def forward(self, x):
        return ReflectPadFn.apply(x, self.padding)


# ----------------------------------------------------------------------------- fusing container
def _is_norm(m):
    return isinstance(m, (InstanceNorm2d, BatchNorm2d))


def conv_norm_act(conv, norm, x, act=ACT_NONE, slope=0.0, residual=None, reflect=0):
    """The reference's fusion unit (arch/ops.py:40-57; Bottleneck conv+bn pairs, arch/generators.py:345-365):
    conv -> norm [+ residual] -> activation.  The norm's batch statistics come out of the conv's epilogue when the library
    can fuse them (sscg_conv2d_fwd_stats); the output is then read once (normalise) instead of twice."""
    spec = norm.stat_spec() if (F.FUSE_STATS[0] and isinstance(conv, Conv2d)) else None
    if spec is None:
        y = conv(x, reflect=reflect) if isinstance(conv, Conv2d) else conv(x)
        return norm(y, act, slope, residual=residual)
    if not ONE_NODE[0]:
        y, stats = conv.forward_stats(x, spec, reflect)
        return norm(y, act, slope, residual=residual, stats=stats)
    # one autograd node for the whole unit (the same kernels; half the host's per-layer cost)
    per_sample, eps, rmean, rvar, momentum = spec
    x, pad, mode = conv._geometry(x, reflect)
    if isinstance(norm, BatchNorm2d):
        norm._pending += _BATCH_GROUPS[0]          # num_batches_tracked, as BatchNorm2d.forward counts it
        gamma, beta = norm.weight, norm.bias
    else:
        gamma = beta = None
    return F.conv_norm_act(x, conv.weight, conv.bias, conv.stride, pad, conv.dilation, mode, gamma, beta, residual, rmean, rvar,
                           per_sample, eps, momentum, act, slope)


FUSE_HEAD = [os.environ.get("SSCG_FUSE_HEAD", "1") != "0"]     # A/B aid: PixelDiscriminator's tail as separate norm / conv launches


def _is_pixel_head(conv, norm, act, head):
    """conv -> norm -> ReLU / LeakyReLU -> Conv2d(C, 1, 1x1): PixelDiscriminator from its second conv on
    (arch/discriminators.py:70-75), served by one fused node when the norm uses batch statistics."""
    return (FUSE_HEAD[0] and ONE_NODE[0] and F.FUSE_STATS[0] and isinstance(conv, Conv2d) and isinstance(head, Conv2d)
            and head.out_channels == 1 and head.kernel_size == 1 and head.stride == 1 and head.padding == 0
            and act.code in (ACT_RELU, ACT_LRELU) and norm.stat_spec() is not None and F.norm_head_applies(conv.out_channels))


def conv_norm_act_head(conv, norm, act, head, x, reflect=0):
    per_sample, eps, rmean, rvar, momentum = norm.stat_spec()
    x, pad, mode = conv._geometry(x, reflect)
    if isinstance(norm, BatchNorm2d):
        norm._pending += _BATCH_GROUPS[0]          # num_batches_tracked, as BatchNorm2d.forward counts it
        gamma, beta = norm.weight, norm.bias
    else:
        gamma = beta = None
    return F.conv_norm_act_head(x, conv.weight, conv.bias, conv.stride, pad, conv.dilation, mode, gamma, beta, head.weight, head.bias,
                                rmean, rvar, per_sample, eps, momentum, act.code, act.slope)


def _is_pixel_front(conv1, act1, conv2, x):
    """Conv2d(cin, 64, 1x1) -> LeakyReLU in front of a PixelDiscriminator tail (arch/discriminators.py:70-71): one launch with the
    tail's conv where the library serves it (fp32 tensors, split mode, cin in {3, 4, 20, 21})."""
    return (isinstance(conv1, Conv2d) and isinstance(act1, _Act) and act1.code == ACT_LRELU and conv1.kernel_size == 1
            and conv1.stride == 1 and conv1.padding == 0 and conv1.dilation == 1 and conv2.kernel_size == 1 and conv2.stride == 1
            and conv2.padding == 0 and conv2.dilation == 1
            and F.conv2d_front_applies(x, conv1.weight, conv2.weight, 1, 0, 1, PAD_ZEROS))


def pixel_disc(conv1, act1, conv, norm, act, head, x):
    per_sample, eps, rmean, rvar, momentum = norm.stat_spec()
    if isinstance(norm, BatchNorm2d):
        norm._pending += _BATCH_GROUPS[0]          # num_batches_tracked, as BatchNorm2d.forward counts it
        gamma, beta = norm.weight, norm.bias
    else:
        gamma = beta = None
    return F.pixel_disc(x, conv1.weight, conv1.bias, act1.slope, conv.weight, conv.bias, gamma, beta, head.weight, head.bias,
                        rmean, rvar, per_sample, eps, momentum, act.code, act.slope)


ONE_NODE = [os.environ.get("SSCG_ONE_NODE", "1") != "0"]       # A/B aid: conv and norm as two autograd nodes


class FusedSequential(nn.Sequential):
    """nn.Sequential whose forward folds [ReflectionPad2d] Conv [Norm] [Activation] [+residual] runs into fused launches."""

    def forward(self, x, reflect=0):
        mods = list(self)
        i, n = 0, len(mods)
        while i < n:
            m = mods[i]
            if isinstance(m, ReflectionPad2d) and i + 1 < n and isinstance(mods[i + 1], (Conv2d, FusedSequential)):
                reflect = m.padding
                i += 1
                continue
            if isinstance(m, FusedSequential):
                x = m(x, reflect=reflect)
                reflect = 0
                i += 1
                continue
            if isinstance(m, (Conv2d, ConvTranspose2d)):
                nxt = mods[i + 1] if i + 1 < n else None
                nx2 = mods[i + 2] if i + 2 < n else None
                if (not reflect and i + 5 < n and isinstance(nx2, Conv2d) and _is_norm(mods[i + 3]) and isinstance(mods[i + 4], _Act)
                        and _is_pixel_head(nx2, mods[i + 3], mods[i + 4], mods[i + 5]) and _is_pixel_front(m, nxt, nx2, x)):
                    x = pixel_disc(m, nxt, nx2, mods[i + 3], mods[i + 4], mods[i + 5], x)      # the whole PixelDiscriminator: one node
                    i += 6
                    continue
                if _is_norm(nxt):
                    nx3 = mods[i + 3] if i + 3 < n else None
                    if isinstance(nx2, _Act) and _is_pixel_head(m, nxt, nx2, nx3):
                        x = conv_norm_act_head(m, nxt, nx2, nx3, x, reflect)
                        i += 4
                    elif isinstance(nx2, _Act):
                        x = conv_norm_act(m, nxt, x, nx2.code, nx2.slope, reflect=reflect)
                        i += 3
                    else:
                        x = conv_norm_act(m, nxt, x, reflect=reflect)
                        i += 2
                elif isinstance(nxt, _Act):
                    x = m(x, reflect, nxt.code, nxt.slope) if isinstance(m, Conv2d) else m(x, nxt.code, nxt.slope)
                    i += 2
                else:
                    x = m(x, reflect=reflect) if isinstance(m, Conv2d) else m(x)
                    i += 1
                reflect = 0
                continue
            if reflect:
                raise SscgError("dangling ReflectionPad2d before %s" % type(m).__name__)
            x = m(x)
            i += 1
        return x


# ----------------------------------------------------------------------------- reference factories
class NormLayer:
    """What get_norm_layer returns: call it with a channel count to get the norm module.
    (.func mirrors functools.partial so reference-style `norm_layer.func == nn.InstanceNorm2d` tests keep working.)"""

    def __init__(self, kind):
        self.kind = kind
        self.func = nn.InstanceNorm2d if kind == "instance" else nn.BatchNorm2d

    def __call__(self, num_features):
        return InstanceNorm2d(num_features) if self.kind == "instance" else BatchNorm2d(num_features)


def as_norm_layer(norm_layer):
    """Accept this package's NormLayer, torch's nn.BatchNorm2d / nn.InstanceNorm2d classes, or a functools.partial of them."""
    if isinstance(norm_layer, NormLayer):
        return norm_layer
    f = norm_layer.func if isinstance(norm_layer, functools.partial) else norm_layer
    if f in (nn.InstanceNorm2d, InstanceNorm2d):
        return NormLayer("instance")
    if f in (nn.BatchNorm2d, BatchNorm2d):
        return NormLayer("batch")
    raise NotImplementedError("normalization layer [%s] is not found" % norm_layer)


def get_norm_layer(norm_type="instance"):
    if norm_type not in ("batch", "instance"):
        raise NotImplementedError("normalization layer [%s] is not found" % norm_type)
    return NormLayer(norm_type)


def init_weights(net, init_type="normal", gain=0.02):
    """arch/ops.py:16-28: Conv/Linear weights ~ N(0, gain), biases 0; BatchNorm2d weight ~ N(1, gain), bias 0."""
    def visit(m):
        cls = m.__class__.__name__
        w = getattr(m, "weight", None)
        if w is not None and (cls.find("Conv") != -1 or cls.find("Linear") != -1):
            with torch.no_grad():
                w.copy_(torch.empty(w.shape).normal_(0.0, gain))
                if getattr(m, "bias", None) is not None:
                    m.bias.zero_()
        elif cls.find("BatchNorm2d") != -1:
            with torch.no_grad():
                m.weight.copy_(torch.empty(m.weight.shape).normal_(1.0, gain))
                m.bias.zero_()

    print("Network initialized with weights sampled from N(0,0.02).")
    net.apply(visit)


def init_network(net, gpu_ids=[]):
    init_weights(net)   # drawn on the host (reproducible across devices), then moved
    if len(gpu_ids) > 0:
        assert torch.cuda.is_available()
        net.cuda(gpu_ids[0])
    return net


def conv_norm_lrelu(in_dim, out_dim, kernel_size, stride=1, padding=0, norm_layer=nn.BatchNorm2d, bias=False):
    return FusedSequential(Conv2d(in_dim, out_dim, kernel_size, stride, padding, bias=bias),
                           as_norm_layer(norm_layer)(out_dim), LeakyReLU(0.2, True))


def conv_norm_relu(in_dim, out_dim, kernel_size, stride=1, padding=0, norm_layer=nn.BatchNorm2d, bias=False):
    return FusedSequential(Conv2d(in_dim, out_dim, kernel_size, stride, padding, bias=bias),
                           as_norm_layer(norm_layer)(out_dim), ReLU(True))


def dconv_norm_relu(in_dim, out_dim, kernel_size, stride=1, padding=0, output_padding=0, norm_layer=nn.BatchNorm2d, bias=False):
    return FusedSequential(ConvTranspose2d(in_dim, out_dim, kernel_size, stride, padding, output_padding, bias=bias),
                           as_norm_layer(norm_layer)(out_dim), ReLU(True))


class ResidualBlock(nn.Module):
    """arch/ops.py:59-74: x + [RefPad1, conv3x3, norm, ReLU, (Dropout .5), RefPad1, conv3x3, norm](x).
    The skip connection is added inside the second normalise pass."""

    def __init__(self, dim, norm_layer, use_dropout, use_bias):
        super().__init__()
        nl = as_norm_layer(norm_layer)
        blk = [ReflectionPad2d(1), conv_norm_relu(dim, dim, kernel_size=3, norm_layer=nl, bias=use_bias)]
        if use_dropout:
            blk += [Dropout(0.5)]
        blk += [ReflectionPad2d(1), Conv2d(dim, dim, 3, padding=0, bias=use_bias), nl(dim)]
        self.res_block = FusedSequential(*blk)

    def forward(self, x):
        mods = list(self.res_block)
        x, shortcut = F.split(x)
        h = mods[1](x, reflect=1)
        k = 2
        if isinstance(mods[k], Dropout):
            h = mods[k](h)
            k += 1
        conv, norm = mods[k + 1], mods[k + 2]
        return conv_norm_act(conv, norm, h, ACT_NONE, 0.0, residual=shortcut, reflect=1)


def set_grad(nets, requires_grad=False):
    for net in nets:
        for param in net.parameters():
            param.requires_grad = requires_grad


import os as _os
if _os.environ.get("SSCG_RACECHECK"):     # debug: see functional.py (stream-ordering checker)
    import sys as _sys
    from .._lib import dev_tool as _dev_tool
    _dev_tool("racecheck").wrap_functions(_sys.modules[__name__])
