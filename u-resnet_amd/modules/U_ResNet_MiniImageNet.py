"""Classification U-ResNet for 224x224 inputs: the runnable classes of the reference's
Classification/modules/U_ResNet_MiniImageNet.py (C/M), with HIP layer bodies.
"""
from .. import nn as hnn
from .. import ops
from ._blocks import (BasicBlock, DoubleConv, Down, EUCB, EUCB2x3, OutConvPlain, ResTrunk, Up,  # noqa: F401
                      act_layer, channel_shuffle, stem)
from .U_ResNet_CIFAR import _Cls

EUCB2 = EUCB2x3
OutConv = OutConvPlain

__all__ = ["channel_shuffle", "act_layer", "EUCB", "EUCB2", "DoubleConv", "Down", "Up", "OutConv", "UNet",
           "BasicBlock", "CombinePooling", "ResNet18", "U_ResNet"]


class UNet(hnn.Module):
    """C/M:162-212: six levels; only up0/up1 are used (up2..up4 are dead parameters kept for the checkpoint)."""

    def __init__(self, in_channels=1, num_classes=2, base_c=64):
        super().__init__()
        self.in_channels = in_channels
        self.num_classes = num_classes
        self.in_conv = DoubleConv(in_channels, base_c)
        self.down1 = Down(base_c, base_c * 2)
        self.down2 = Down(base_c * 2, base_c * 4)
        self.down3 = Down(base_c * 4, base_c * 8)
        self.down4 = Down(base_c * 8, base_c * 16)
        self.down5 = Down(base_c * 16, base_c * 32)
        self.up0 = Up(base_c * 32, base_c * 16)
        self.up1 = Up(base_c * 16, base_c * 8)
        self.up2 = Up(base_c * 8, base_c * 4)
        self.up3 = Up(base_c * 4, base_c * 2)
        self.up4 = Up(base_c * 2, base_c)
        self.out_conv = OutConv(base_c * 8, num_classes)

    def forward(self, x):
        x1 = self.in_conv(x)
        x2 = self.down1(x1)
        x3 = self.down2(x2)
        x4 = self.down3(x3)
        x5 = self.down4(x4)
        x6 = self.down5(x5)
        x = self.up0(x6, x5)
        x = self.up1(x, x4)
        return self.out_conv(x)


class CombinePooling(hnn.Module):
    """C/M:306-319: constructed by U_ResNet for its parameters, never called by any forward."""

    def __init__(self, kernel_size, stride):
        super().__init__()
        self.Conv = hnn.Conv2d(kernel_size=1, stride=1, in_channels=712, out_channels=712 // 2)

    def forward(self, x):
        raise NotImplementedError("CombinePooling is dead code in the reference (never called); no HIP kernel")


class ResNet18(_Cls):
    """C/M:269-304."""

    def __init__(self, ResidulBlock, num_classes=10):
        super().__init__()
        self.name = "ResNet18"
        self.compute_dtype = hnn.get_default_dtype()
        self.conv1 = stem(3)
        self.inchannel = 64
        self.layer1 = self.make_layer(ResidulBlock, 64, 2, stride=1)
        self.layer2 = self.make_layer(ResidulBlock, 128, 2, stride=2)
        self.layer3 = self.make_layer(ResidulBlock, 256, 2, stride=2)
        self.layer4 = self.make_layer(ResidulBlock, 512, 2, stride=2)
        self.fc = hnn.Linear(512 * 4, num_classes)

    def forward(self, x):
        from ._blocks import run_stem
        x = ops.as_activation(x, self.compute_dtype)
        out = run_stem(self.conv1, x)
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            out = layer(out)
        return self._head(out, 7)


class U_ResNet(_Cls):
    """C/M:321-365."""

    def __init__(self, num_classes, in_channels, ResidulBlock=BasicBlock):
        super().__init__()
        self.name = "U_ResNet"
        self.compute_dtype = hnn.get_default_dtype()
        self.conv1 = stem(in_channels)
        self.U_Net = UNet(in_channels=3, num_classes=num_classes, base_c=64)
        self._build_res(ResidulBlock, (64, 128, 256, 512), (1, 2, 2, 2))
        self.Upsample = EUCB2(in_channels=512, out_channels=256, scale_factor=2)
        self.CBPooling = CombinePooling(kernel_size=7, stride=7)
        self.fc = hnn.Linear(5696, num_classes)

    def forward(self, x):
        x = ops.as_activation(x, self.compute_dtype)
        out, logits = ops.two_branches(x, lambda t: self.Upsample(self._res_forward(t)), self.U_Net)
        return self._head(ops.cat_channels(out, logits), 7)
