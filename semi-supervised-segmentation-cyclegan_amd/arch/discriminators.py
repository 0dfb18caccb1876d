"""Discriminators of the reference (arch/discriminators.py), evaluated by libsscg.so kernels.

state_dict keys match the reference: Pixel `dis_model.{0,2,5}.*`, NLayer `dis_model.0.*, dis_model.{2,3,4}.0.*,
dis_model.5.*`, FC `conv{1..4}.*, classifier.*`."""
from torch import nn

from .._lib import ACT_LRELU
from .ops import (Conv2d, FusedSequential, LeakyReLU, as_norm_layer, conv_norm_lrelu, get_norm_layer, init_network)


class PixelDiscriminator(nn.Module):
    """1x1 conv(in->ndf)+LReLU -> 1x1 conv(ndf->2ndf) -> norm -> LReLU -> 1x1 conv(2ndf->1)   (arch/discriminators.py:66-80).
    The trained Di / Ds and the frozen old_Di of the training step (model.py:220-230)."""

    def __init__(self, input_nc, ndf=64, norm_layer=nn.BatchNorm2d, use_bias=False):
        super().__init__()
        nl = as_norm_layer(norm_layer)
        self.dis_model = FusedSequential(
            Conv2d(input_nc, ndf, 1, 1, 0), LeakyReLU(0.2, True),
            Conv2d(ndf, ndf * 2, 1, 1, 0, bias=use_bias), nl(ndf * 2), LeakyReLU(0.2, True),
            Conv2d(ndf * 2, 1, 1, 1, 0, bias=use_bias))
        self.dis_model[-1].head = True

    def forward(self, input):
        return self.dis_model(input)


class NLayerDiscriminator(nn.Module):
    """70x70 PatchGAN (arch/discriminators.py:42-63)."""

    def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer=nn.BatchNorm2d, use_bias=False):
        super().__init__()
        nl = as_norm_layer(norm_layer)
        layers = [Conv2d(input_nc, ndf, 4, 2, 1), LeakyReLU(0.2, True)]
        mult = 1
        for n in range(1, n_layers):
            prev, mult = mult, min(2 ** n, 8)
            layers.append(conv_norm_lrelu(ndf * prev, ndf * mult, 4, 2, 1, norm_layer=nl, bias=use_bias))
        prev, mult = mult, min(2 ** n_layers, 8)
        layers.append(conv_norm_lrelu(ndf * prev, ndf * mult, 4, 1, 1, norm_layer=nl, bias=use_bias))
        layers.append(Conv2d(ndf * mult, 1, 4, 1, 1))
        layers[-1].head = True
        self.dis_model = FusedSequential(*layers)

    def forward(self, input):
        return self.dis_model(input)


class FCDiscriminator(nn.Module):
    """Five 4x4 stride-2 convs with LeakyReLU(0.2) between them (arch/discriminators.py:8-39)."""

    def __init__(self, num_classes, ndf=64):
        super().__init__()
        self.conv1 = Conv2d(num_classes, ndf, 4, 2, 1)
        self.conv2 = Conv2d(ndf, ndf * 2, 4, 2, 1)
        self.conv3 = Conv2d(ndf * 2, ndf * 4, 4, 2, 1)
        self.conv4 = Conv2d(ndf * 4, ndf * 8, 4, 2, 1)
        self.classifier = Conv2d(ndf * 8, 1, 4, 2, 1)
        self.classifier.head = True

    def forward(self, x):
        for conv in (self.conv1, self.conv2, self.conv3, self.conv4):
            x = conv(x, 0, ACT_LRELU, 0.2)
        return self.classifier(x)


def define_Dis(input_nc, ndf, netD, n_layers_D=3, norm='batch', gpu_ids=[0]):
    nl = get_norm_layer(norm_type=norm)
    use_bias = nl.kind == "instance"
    if netD == 'n_layers':
        net = NLayerDiscriminator(input_nc, ndf, n_layers_D, norm_layer=nl, use_bias=use_bias)
    elif netD == 'pixel':
        net = PixelDiscriminator(input_nc, ndf, norm_layer=nl, use_bias=use_bias)
    elif netD == 'fc_disc':
        net = FCDiscriminator(input_nc, ndf)
    else:
        raise NotImplementedError('Discriminator model name [%s] is not recognized' % netD)
    return init_network(net, gpu_ids)
