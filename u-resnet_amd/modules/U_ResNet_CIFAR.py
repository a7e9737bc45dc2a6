"""Classification U-ResNet family for 32x32 inputs: the classes of the reference's
Classification/modules/U_ResNet_CIFAR.py (C/C) that run as committed, with HIP layer bodies.

`U_ResNet`, `S_ResNet` and `NlNet` (C/C:296, 342, 389) are shape-broken at 32x32 as the reference
ships them (hard-coded `fc` sizes, a concatenation of 8x8 with 16x16 maps): they are constructed
with the same signatures, parameters and state_dict keys, run the same layers, and fail in
`forward` with the RuntimeError text the reference produces.
Outputs are [B, num_classes] in the compute dtype.
"""
from .. import nn as hnn
from .. import ops
from ._blocks import (BasicBlock, DoubleConv, Down, EUCB, EUCB2x3, OutConvPlain, ResTrunk, Up,  # noqa: F401
                      act_layer, add_unet_branch, channel_shuffle, stem)

EUCB2 = EUCB2x3          # C/C:64-100: three grouped stages
OutConv = OutConvPlain   # C/C:148-152

__all__ = ["channel_shuffle", "act_layer", "EUCB", "EUCB2", "DoubleConv", "Down", "Up", "OutConv", "UNet",
           "BasicBlock", "E_ResNet", "U_ResNet", "S_ResNet", "NlNet", "UResNet3_No_EUCB", "UResNet3_No_UBlock",
           "UResNet4_No_SU"]


class UNet(hnn.Module):
    """C/C:155-211: up4 is constructed but never called; the head always has 100 output channels."""

    def __init__(self, in_channels=1, num_classes=2, base_c=64):
        super().__init__()
        self.in_channels = in_channels
        self.num_classes = num_classes
        add_unet_branch(self, in_channels, base_c)
        self.up1 = Up(base_c * 16, base_c * 8)
        self.up2 = Up(base_c * 8, base_c * 4)
        self.up3 = Up(base_c * 4, base_c * 2)
        self.up4 = Up(base_c * 2, base_c)
        self.out_conv = OutConv(base_c * 2, 100)

    def forward(self, x):
        x1 = self.in_conv(x)
        x2 = self.down1(x1)
        x3 = self.down2(x2)
        x4 = self.down3(x3)
        x5 = self.down4(x4)
        x = self.up1(x5, x4)
        x = self.up2(x, x3)
        x = self.up3(x, x2)
        return self.out_conv(x)


class _Cls(ResTrunk):
    def _head(self, feat, k):
        out = ops.avg_pool2d(feat, k)
        out = ops.to_nchw(out).view(out.size(0), -1)       # the reference flattens in (C, H, W) order
        return self.fc(out)


class UResNet4_No_SU(_Cls):
    """C/C:533-578 (instantiated by Classification/CNN-CIFAR10.py:121 with scale=3)."""

    def __init__(self, num_classes, in_channels, ResidulBlock=BasicBlock, scale=2):
        super().__init__()
        self.name = "UResNet4_No_SU"
        self.compute_dtype = hnn.get_default_dtype()
        self.conv1 = stem(in_channels)
        self.U_Net = UNet(in_channels=in_channels, num_classes=num_classes, base_c=64)
        self._build_res(ResidulBlock, (64, 128, 256, 512), (1, 2, 2, 1))
        self.Upsample = EUCB2(in_channels=512, out_channels=256, scale_factor=scale)
        self.fc = hnn.Linear(384, num_classes)
        self.convert = hnn.Conv2d(kernel_size=5, in_channels=100, out_channels=128)

    def forward(self, x):
        x = ops.as_activation(x, self.compute_dtype)
        out, logits = ops.two_branches(x, lambda t: self.Upsample(self._res_forward(t)), lambda t: self.convert(self.U_Net(t)))
        return self._head(ops.cat_channels(out, logits), 7)


class UResNet3_No_EUCB(_Cls):
    """C/C:437-485."""

    def __init__(self, num_classes, in_channels, ResidulBlock=BasicBlock, scale=2):
        super().__init__()
        self.name = "U-ResNet3_No_EUCB"
        self.scale = scale
        self.compute_dtype = hnn.get_default_dtype()
        self.conv1 = stem(in_channels, selected_upsampling=True)
        self.U_Net = UNet(in_channels=in_channels, num_classes=num_classes, base_c=64)
        self._build_res(ResidulBlock, (64, 128, 256, 512), (1, 2, 2, 1))
        self.fc = hnn.Linear(612, num_classes)
        self.convert = hnn.Conv2d(kernel_size=5, in_channels=100, out_channels=128)

    def forward(self, x):
        x = ops.as_activation(x, self.compute_dtype)
        out, logits = ops.two_branches(x, self._res_forward, lambda t: ops.avg_pool2d(self.U_Net(t), 2))
        return self._head(ops.cat_channels(out, logits), 5)


class UResNet3_No_UBlock(_Cls):
    """C/C:487-531 (U_Net is constructed but unused)."""

    def __init__(self, num_classes, in_channels, ResidulBlock=BasicBlock, scale=2):
        super().__init__()
        self.name = "UResNet3_No_UBlock"
        self.scale = scale
        self.compute_dtype = hnn.get_default_dtype()
        self.conv1 = stem(in_channels, selected_upsampling=True)
        self.U_Net = UNet(in_channels=in_channels, num_classes=num_classes, base_c=64)
        self._build_res(ResidulBlock, (64, 128, 256, 512), (1, 2, 2, 2))
        self.Upsample = EUCB2(in_channels=512, out_channels=256 * 2, scale_factor=scale)
        self.fc = hnn.Linear(512, num_classes)
        self.convert = hnn.Conv2d(kernel_size=5, in_channels=100, out_channels=128)

    def forward(self, x):
        x = ops.as_activation(x, self.compute_dtype)
        return self._head(self.Upsample(self._res_forward(x)), 7)


class E_ResNet(_Cls):
    """C/C:252-294."""

    def __init__(self, num_classes=10, in_channels=3, ResidulBlock=BasicBlock):
        super().__init__()
        self.name = "E_ResNet"
        self.compute_dtype = hnn.get_default_dtype()
        self.conv1 = stem(in_channels, selected_upsampling=True)
        self._build_res(ResidulBlock, (64, 128, 256, 512), (1, 2, 2, 1))
        self.Upsample = EUCB(in_channels=512, out_channels=256 * 2, scale_factor=2)
        self.fc = hnn.Linear(2048, num_classes)

    def forward(self, x):
        x = ops.as_activation(x, self.compute_dtype)
        return self._head(self.Upsample(self._res_forward(x)), 7)


class U_ResNet(_Cls):
    """C/C:296-340.  Its U-Net always takes 3 input channels (:308); at 32x32 the merged map has 612 channels against
    fc's 522 (:313), so forward raises as the reference's does."""

    def __init__(self, num_classes=10, in_channels=3, ResidulBlock=BasicBlock):
        super().__init__()
        self.name = "U_ResNet"
        self.compute_dtype = hnn.get_default_dtype()
        self.conv1 = stem(in_channels, selected_upsampling=True)
        self.U_Net = UNet(in_channels=3, num_classes=num_classes, base_c=64)
        self._build_res(ResidulBlock, (64, 128, 256, 512), (1, 2, 2, 1))
        self.fc = hnn.Linear(522, num_classes)

    def forward(self, x):
        x = ops.as_activation(x, self.compute_dtype)
        logits = ops.avg_pool2d(self.U_Net(x), 2)
        out = self._res_forward(x)
        return self._head(ops.cat_channels(out, logits), 7)


class S_ResNet(_Cls):
    """C/C:342-387: no selected upsampling, EUCB after the trunk; same fc mismatch as U_ResNet at 32x32 (:359)."""

    def __init__(self, num_classes=10, in_channels=3, ResidulBlock=BasicBlock):
        super().__init__()
        self.name = "S_ResNet"
        self.compute_dtype = hnn.get_default_dtype()
        self.conv1 = stem(in_channels)
        self.U_Net = UNet(in_channels=3, num_classes=num_classes, base_c=64)
        self._build_res(ResidulBlock, (64, 128, 256, 512), (1, 2, 2, 1))
        self.Upsample = EUCB(in_channels=512, out_channels=256 * 2, scale_factor=2)
        self.fc = hnn.Linear(522, num_classes)

    def forward(self, x):
        x = ops.as_activation(x, self.compute_dtype)
        logits = ops.avg_pool2d(self.U_Net(x), 2)
        out = self.Upsample(self._res_forward(x))
        return self._head(ops.cat_channels(out, logits), 7)


class NlNet(_Cls):
    """C/C:389-435: three strided stages and EUCB* leave an 8x8 map (scale=2) that is concatenated with the U-Net's
    16x16 logits (:427): forward raises torch.cat's size error, as in the reference.  The shape print of :426 is not
    reproduced."""

    def __init__(self, num_classes, in_channels, ResidulBlock=BasicBlock, scale=2):
        super().__init__()
        self.name = "NlNet"
        self.compute_dtype = hnn.get_default_dtype()
        self.conv1 = stem(in_channels, selected_upsampling=True)
        self.U_Net = UNet(in_channels=in_channels, num_classes=num_classes, base_c=64)
        self._build_res(ResidulBlock, (64, 128, 256, 512), (2, 2, 2, 1))
        self.Upsample = EUCB2(in_channels=512, out_channels=256, scale_factor=scale)
        self.fc = hnn.Linear(1424, num_classes)
        self.convert = hnn.Conv2d(kernel_size=5, in_channels=100, out_channels=128)

    def forward(self, x):
        x = ops.as_activation(x, self.compute_dtype)
        logits = self.U_Net(x)
        out = self.Upsample(self._res_forward(x))
        return self._head(ops.cat_channels(out, logits), 7)
