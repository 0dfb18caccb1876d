"""Generators of the reference's hot path (arch/generators.py), evaluated by libsscg.so kernels.

In scope: `deeplab` (ResNet-101, output stride 8 - the trained Gis/Gsi, arch/generators.py:320-441)
and `resnet_{6,9}blocks[_softmax]` (the classic CycleGAN generator - the frozen old_Gis/old_Gsi, :65-95).
`unet_128` / `unet_256` (:7-63; SURVEY 8(f) N4: reachable through `--gen_net` with `--honour_nets 1`) run on the same kernels.
`enet`, `lednet_*` are never constructed by any driver of the reference and are out of scope; asking
for them raises.  state_dict keys equal the reference's (checkpoint ABI)."""
from torch import nn

from .. import functional as F
from .._lib import ACT_RELU
from .ops import (BatchNorm2d, Conv2d, ConvTranspose2d, Dropout, FusedSequential, LeakyReLU, ReLU, ReflectionPad2d, ResidualBlock,
                  Tanh, as_norm_layer, conv_norm_act, conv_norm_relu, dconv_norm_relu, get_norm_layer, init_network)


class ResnetGenerator(nn.Module):
    """RefPad3, c7s1-ngf, d2ngf, d4ngf, num_blocks x R4ngf, u2ngf, u-ngf, RefPad3, c7s1-out [, tanh]."""

    def __init__(self, input_nc=3, output_nc=3, ngf=64, norm_layer=nn.BatchNorm2d, use_dropout=True, num_blocks=6, softmax=False):
        super().__init__()
        nl = as_norm_layer(norm_layer)
        bias = nl.kind == "instance"
        seq = [ReflectionPad2d(3),
               conv_norm_relu(input_nc, ngf, 7, norm_layer=nl, bias=bias),
               conv_norm_relu(ngf, ngf * 2, 3, 2, 1, norm_layer=nl, bias=bias),
               conv_norm_relu(ngf * 2, ngf * 4, 3, 2, 1, norm_layer=nl, bias=bias)]
        seq += [ResidualBlock(ngf * 4, nl, use_dropout, bias) for _ in range(num_blocks)]
        seq += [dconv_norm_relu(ngf * 4, ngf * 2, 3, 2, 1, 1, norm_layer=nl, bias=bias),
                dconv_norm_relu(ngf * 2, ngf, 3, 2, 1, 1, norm_layer=nl, bias=bias),
                ReflectionPad2d(3),
                Conv2d(ngf, output_nc, 7)]
        seq[-1].head = True
        if not softmax:   # softmax=True only drops the Tanh; no softmax layer is added (arch/generators.py:81-91)
            seq.append(Tanh())
        self.res_model = FusedSequential(*seq)

    def forward(self, x):
        return self.res_model(x)


class UnetSkipConnectionBlock(nn.Module):
    """arch/generators.py:7-45.  x -> cat([x, up(submodule(down(x)))], 1); the outermost block returns up(...) alone.

    The reference's LeakyReLU(0.2, True) at the head of `down` works IN PLACE on x, so the skip branch of every inner block
    carries the ACTIVATED x (the well-known pix2pix behaviour): restated here explicitly (the activation runs once, both
    branches read its result)."""

    def __init__(self, outer_nc, inner_nc, input_nc=None, submodule=None, outermost=False, innermost=False,
                 norm_layer=nn.BatchNorm2d, use_dropout=False):
        super().__init__()
        self.outermost = outermost
        nl = as_norm_layer(norm_layer)
        use_bias = nl.kind == "instance"
        if input_nc is None:
            input_nc = outer_nc
        downconv = Conv2d(input_nc, inner_nc, 4, 2, 1, bias=use_bias)
        if outermost:
            upconv = ConvTranspose2d(inner_nc * 2, outer_nc, 4, 2, 1)
            upconv.head = True
            model = [downconv, submodule, ReLU(True), upconv]
        elif innermost:
            upconv = ConvTranspose2d(inner_nc, outer_nc, 4, 2, 1, bias=use_bias)
            model = [LeakyReLU(0.2, True), downconv, ReLU(True), upconv, nl(outer_nc)]
        else:
            upconv = ConvTranspose2d(inner_nc * 2, outer_nc, 4, 2, 1, bias=use_bias)
            model = [LeakyReLU(0.2, True), downconv, nl(inner_nc), submodule, ReLU(True), upconv, nl(outer_nc)]
            if use_dropout:
                model.append(Dropout(0.5))
        self.model = FusedSequential(*model)
        # everything behind the leading LeakyReLU, as a fused run (not a registered child: the state_dict keys stay the reference's)
        self.__dict__["_tail"] = None if outermost else FusedSequential(*model[1:])

    def forward(self, x):
        if self.outermost:
            return self.model(x)
        xa = self.model[0](x)                    # the in-place LeakyReLU: both branches see the activated tensor
        xa, xs = F.split(xa, 2)                  # explicit fan-out: the gradient sum runs in sscg_add
        return F.cat_channels(xs, self._tail(xa))


class UnetGenerator(nn.Module):
    """arch/generators.py:47-63: num_downs stride-2 4x4 convs down to the bottleneck and back (unet_128: 7, unet_256: 8)."""

    def __init__(self, input_nc, output_nc, num_downs, ngf=64, norm_layer=nn.BatchNorm2d, use_dropout=False):
        super().__init__()
        nl = as_norm_layer(norm_layer)
        block = UnetSkipConnectionBlock(ngf * 8, ngf * 8, submodule=None, norm_layer=nl, innermost=True)
        for _ in range(num_downs - 5):
            block = UnetSkipConnectionBlock(ngf * 8, ngf * 8, submodule=block, norm_layer=nl, use_dropout=use_dropout)
        block = UnetSkipConnectionBlock(ngf * 4, ngf * 8, submodule=block, norm_layer=nl)
        block = UnetSkipConnectionBlock(ngf * 2, ngf * 4, submodule=block, norm_layer=nl)
        block = UnetSkipConnectionBlock(ngf, ngf * 2, submodule=block, norm_layer=nl)
        self.unet_model = UnetSkipConnectionBlock(output_nc, ngf, input_nc=input_nc, submodule=block, outermost=True, norm_layer=nl)

    def forward(self, x):
        return self.unet_model(x)


class Bottleneck(nn.Module):
    """1x1(stride) -> BN -> ReLU -> 3x3(dilation) -> BN -> ReLU -> 1x1(x4) -> BN (+ shortcut) -> ReLU
    (arch/generators.py:320-365).  BN affine parameters are frozen; the shortcut add and the last ReLU are
    fused into bn3's normalise pass."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None):
        super().__init__()
        self.conv1 = Conv2d(inplanes, planes, 1, stride, bias=False)
        self.bn1 = BatchNorm2d(planes)
        self.conv2 = Conv2d(planes, planes, 3, 1, dilation, dilation, bias=False)
        self.bn2 = BatchNorm2d(planes)
        self.conv3 = Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = BatchNorm2d(planes * 4)
        for bn in (self.bn1, self.bn2, self.bn3):
            for p in bn.parameters():
                p.requires_grad = False
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        x, shortcut = F.split(x)           # the block input feeds conv1 and the shortcut: their gradients meet in sscg_add
        out = conv_norm_act(self.conv1, self.bn1, x, ACT_RELU)
        out = conv_norm_act(self.conv2, self.bn2, out, ACT_RELU)
        res = shortcut if self.downsample is None else self.downsample(shortcut)
        return conv_norm_act(self.conv3, self.bn3, out, ACT_RELU, 0.0, residual=res)


class Classifier_Module(nn.Module):
    """Four dilated 3x3 heads are created, the forward returns after adding the second
    (arch/generators.py:367-382): heads 2 and 3 exist in the state dict and never receive a gradient."""

    def __init__(self, dilation_series, padding_series, num_classes):
        super().__init__()
        self.conv2d_list = nn.ModuleList(
            [Conv2d(2048, num_classes, 3, 1, p, d, bias=True) for d, p in zip(dilation_series, padding_series)])
        for conv in self.conv2d_list:
            conv.head = True          # the network's output: fp32 also in bf16 mode

    def forward(self, x):
        a, b = F.split(x)
        return F.AddFn.apply(self.conv2d_list[0](a), self.conv2d_list[1](b))


class ResNet(nn.Module):
    """DeepLab-v2 ResNet trunk (arch/generators.py:384-441): output stride 8 (256 -> 33)."""

    def __init__(self, in_channels, block, layers, num_classes):
        super().__init__()
        self.inplanes = 64
        self.conv1 = Conv2d(in_channels, 64, 7, 2, 3, bias=False)
        self.bn1 = BatchNorm2d(64)
        for p in self.bn1.parameters():
            p.requires_grad = False
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=1, dilation=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=1, dilation=4)
        self.layer5 = Classifier_Module([6, 12, 18, 24], [6, 12, 18, 24], num_classes)

    def _make_layer(self, block, planes, blocks, stride=1, dilation=1):
        # every stage opens with a projection shortcut (arch/generators.py:410-416)
        down = FusedSequential(Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                               BatchNorm2d(planes * block.expansion))
        for p in down[1].parameters():
            p.requires_grad = False
        stage = [block(self.inplanes, planes, stride, dilation=dilation, downsample=down)]
        self.inplanes = planes * block.expansion
        stage += [block(self.inplanes, planes, dilation=dilation) for _ in range(1, blocks)]
        return nn.Sequential(*stage)

    def stem(self, x):
        x = conv_norm_act(self.conv1, self.bn1, x, ACT_RELU)
        return F.MaxPoolFn.apply(x)      # MaxPool2d(3, 2, 1, ceil_mode=True), arch/generators.py:394

    def forward(self, x):
        x = self.stem(x)
        for stage in (self.layer1, self.layer2, self.layer3, self.layer4):
            x = stage(x)
        return self.layer5(x)


_OUT_OF_SCOPE = ("enet", "lednet_128", "lednet_256")


def define_Gen(input_nc, output_nc, ngf, netG, norm='batch', use_dropout=False, gpu_ids=[0]):
    nl = get_norm_layer(norm_type=norm)
    if netG in ('resnet_9blocks', 'resnet_9blocks_softmax', 'resnet_6blocks', 'resnet_6blocks_softmax'):
        net = ResnetGenerator(input_nc, output_nc, ngf, norm_layer=nl, use_dropout=use_dropout,
                              num_blocks=9 if '9blocks' in netG else 6, softmax=netG.endswith('_softmax'))
    elif netG in ('unet_128', 'unet_256'):
        net = UnetGenerator(input_nc, output_nc, 7 if netG == 'unet_128' else 8, ngf, norm_layer=nl, use_dropout=use_dropout)
    elif netG == 'deeplab':
        net = ResNet(in_channels=input_nc, block=Bottleneck, layers=[3, 4, 23, 3], num_classes=output_nc)
    elif netG in _OUT_OF_SCOPE:
        raise NotImplementedError('Generator [%s] is outside the MI355X hot path (never built by the reference drivers)' % netG)
    else:
        raise NotImplementedError('Generator model name [%s] is not recognized' % netG)
    return init_network(net, gpu_ids)
