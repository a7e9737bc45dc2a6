"""The runnable scratch variants of the reference's Classification/modules/temp.py (C/T), with HIP layer bodies.

`U_ResNetNew` (C/T:388-439) is not provided: its second `ResLayer4` assignment leaves a 1024-channel trunk in front
of a 512-channel EUCB*, so the reference class cannot run a forward pass as committed.
Outputs are [B, num_classes] in the compute dtype.
"""
from .. import nn as hnn
from .. import ops
from ._blocks import (BasicBlock, Bottleneck, DoubleConv, Down, EUCB, EUCB2x3, OutConvPlain, Up,  # noqa: F401
                      act_layer, channel_shuffle, run_stem, stem)
from .U_ResNet_CIFAR import _Cls

EUCB2 = EUCB2x3          # C/T:63-99: three grouped stages
OutConv = OutConvPlain   # C/T:154-158

__all__ = ["channel_shuffle", "act_layer", "EUCB", "EUCB2", "DoubleConv", "Down", "Up", "OutConv", "UNet", "UNet2",
           "BasicBlock", "Bottleneck", "NlNet", "E_ResNet", "U_ResNet", "UResNet2"]


class UNet(hnn.Module):
    """C/T:161-216: five Down stages, two Up stages (logits at 1/8 resolution)."""

    _levels, _ups = 5, 2

    def __init__(self, in_channels=1, num_classes=2, base_c=32):
        super().__init__()
        self.in_channels = in_channels
        self.num_classes = num_classes
        self.in_conv = DoubleConv(in_channels, base_c)
        for i in range(self._levels):
            setattr(self, 'down%d' % (i + 1), Down(base_c << i, base_c << (i + 1)))
        for i in range(self._ups):
            setattr(self, 'up%d' % i, Up(base_c << (self._levels - i), base_c << (self._levels - 1 - i)))
        self.out_conv = OutConv(base_c * 8, num_classes)

    def forward(self, x):
        feats = [self.in_conv(x)]
        for i in range(self._levels):
            feats.append(getattr(self, 'down%d' % (i + 1))(feats[-1]))
        u = feats[-1]
        for i in range(self._ups):
            u = getattr(self, 'up%d' % i)(u, feats[-2 - i])
        return self.out_conv(u)


class UNet2(UNet):
    """C/T:482-517: six Down stages, three Up stages; at 224x224 the 3x3 bottleneck comes back as 6x6 and is
    zero-padded to the 7x7 skip (C/T:142-146)."""

    _levels, _ups = 6, 3


class _Temp(_Cls):
    def _build(self, in_channels, unet, block, widths, strides):
        self.compute_dtype = hnn.get_default_dtype()
        self.inchannel = 64
        self.conv1 = stem(in_channels)
        if unet is not None:
            self.U_Net = unet()            # constructed after conv1, as in the reference (seeded-init order)
        for i, (c, s) in enumerate(zip(widths, strides), 1):
            setattr(self, 'ResLayer%d' % i, self.make_layer(block, c, 2, s))
        self._nres = len(widths)

    def _trunk(self, x):
        out = run_stem(self.conv1, x)
        for i in range(1, self._nres + 1):
            out = getattr(self, 'ResLayer%d' % i)(out)
        return out


class NlNet(_Temp):
    """C/T:286-337, for 224x224 inputs."""

    _unet = UNet

    def __init__(self, num_classes=100, in_channels=3, ResidulBlock=BasicBlock):
        super().__init__()
        self.name = "NlNet"
        self._build(in_channels, lambda: self._unet(in_channels=3, num_classes=num_classes, base_c=32), ResidulBlock,
                    (64, 128, 256, 512, 256), (1, 2, 2, 2, 1))
        self.Upsample = EUCB2(in_channels=256, out_channels=256 // 2, scale_factor=2)
        self.fc = hnn.Linear(2052, num_classes)

    def forward(self, x):
        x = ops.as_activation(x, self.compute_dtype)
        out, logits = ops.two_branches(x, lambda t: self.Upsample(self._trunk(t)), self.U_Net)
        return self._head(ops.cat_channels(out, logits), 9)      # AvgPool2d(kernel_size=9, stride=9), C/T:304,331


class UResNet2(NlNet):
    """C/T:519-569."""

    _unet = UNet2

    def __init__(self, num_classes=100, in_channels=3, ResidulBlock=BasicBlock):
        super().__init__(num_classes, in_channels, ResidulBlock)
        self.name = "UResNet2"


class E_ResNet(_Temp):
    """C/T:340-386, for 224x224 inputs."""

    def __init__(self, num_classes, in_channels=3, ResidulBlock=BasicBlock):
        super().__init__()
        self.name = "U_ResNet"
        self._build(in_channels, lambda: UNet(in_channels=3, num_classes=num_classes, base_c=64), ResidulBlock,
                    (64, 128, 256, 512), (1, 2, 2, 2))
        self.fc = hnn.Linear(2448, num_classes)

    def forward(self, x):
        x = ops.as_activation(x, self.compute_dtype)
        out, logits = ops.two_branches(x, self._trunk, lambda t: ops.avg_pool2d(self.U_Net(t), 2))
        return self._head(ops.cat_channels(out, logits), 7)


class U_ResNet(_Temp):
    """C/T:441-481, for 224x224 inputs (fc expects 256 x 4 x 4)."""

    def __init__(self, num_classes, in_channels=3, ResidulBlock=BasicBlock):
        super().__init__()
        self.name = "E_ResNet"
        self._build(in_channels, None, ResidulBlock, (64, 128, 256, 512), (1, 2, 2, 2))
        self.Upsample = EUCB(in_channels=512, out_channels=512 // 2, scale_factor=2)
        self.fc = hnn.Linear(4096, num_classes)

    def forward(self, x):
        x = ops.as_activation(x, self.compute_dtype)
        return self._head(self.Upsample(self._trunk(x)), 7)
