"""Segmentation U-ResNet family: same class names, constructor arguments, sub-module names and
state_dict keys as the reference's Segmentation/Cityscapes/modules/U_ResNet.py (S/C) and
Segmentation/Ice-Seg/modules/U_ResNet.py (S/I), with every layer body running on HIP kernels.

forward(x): x is [B, in_channels, H, W] (NCHW fp32 as the reference's loaders produce, or an NHWC
tensor already in the compute dtype); returns logits of logical shape [B, seg_classes, H, W] in
channels-last memory and the model's compute dtype (`set_compute_dtype`, default
uresnet_amd.get_default_dtype()).
"""
import contextlib

from .. import nn as hnn
from .. import ops
from ._blocks import (BasicBlock, Bottleneck, DoubleConv, Down, EUCB, EUCB2, OutConv, ResTrunk, Up,  # noqa: F401
                      act_layer, add_unet_branch, channel_shuffle, has_hooks, stem, unet_branch_forward)

__all__ = ["channel_shuffle", "act_layer", "EUCB", "EUCB2", "DoubleConv", "Down", "Up", "OutConv", "BasicBlock",
           "Bottleneck", "UNet", "UResNet", "UResNetNewEUCB", "UResNetNoEUCB", "UResNetNoResBlock"]


def _add_ups(m, base_c):
    m.up1 = Up(base_c * 16, base_c * 8)
    m.up2 = Up(base_c * 8, base_c * 4)
    m.up3 = Up(base_c * 4, base_c * 2)
    m.up4 = Up(base_c * 2, base_c)


class _TwoBranch(ResTrunk):
    """Feature Merge of the ResBlock branch (half resolution, then upsampled) and the UBlock branch."""
    _res4 = 32

    def __init__(self, seg_classes=1, in_channels=3, base_c=32):
        super().__init__()
        self.compute_dtype = hnn.get_default_dtype()
        self.conv1 = stem(in_channels)
        self._build_res(BasicBlock, (64, 128, 64, self._res4), (1, 1, 1, 1))
        self.Upsample = self._upsample()
        add_unet_branch(self, in_channels, base_c)
        _add_ups(self, base_c)
        self.out_conv = OutConv(base_c + 16, seg_classes)

    def forward(self, x):
        x = ops.as_activation(x, self.compute_dtype)
        # the two branches only meet at the Feature Merge: with ops.branch_streams the ResBlock branch is launched on a
        # second stream, so its kernels fill the compute units the UBlock branch's kernels leave idle (and vice versa)
        fork = ops.BranchFork(x) if (ops.branch_streams and x.is_cuda) else None
        on_side = fork.branch if fork is not None else contextlib.nullcontext
        with on_side():
            res = self._res_forward(x)
        n, _, h, w = x.shape
        c_u, c_res = self.up4.conv[3].out_channels, self.out_conv[0].num_features - self.up4.conv[3].out_channels
        head_conv = self.out_conv[2]
        if (res.shape[2] * 2 == h and res.shape[3] * 2 == w and not has_hooks(self.out_conv, *self.out_conv)
                and head_conv.stride == (1, 1) and head_conv.padding == (0, 0) and head_conv.groups == 1
                and head_conv.dilation == (1, 1)
                and ops.bn_relu_conv1x1_pair_supported(n, c_u, c_res, h, w, x.dtype, self.out_conv[0], head_conv)):
            # Feature Merge without a merged tensor (S/C:280 torch.cat([u_out, res_out], dim=1) -> S/C:344 out_conv): both
            # branches keep their dense results and the head's kernels read the pair.  (Channel windows of one 80-channel
            # tensor are 128- and 32-byte runs at a 160-byte pitch: partial memory lines in every pass that touches them.)
            with on_side():
                res_out = self.Upsample(res)
            u_out = unet_branch_forward(self, x, out_stats=self.out_conv[0].training)   # (the head's BatchNorm comes next)
            if fork is not None:
                fork.join(res_out)
            z = ops.bn_relu_conv1x1_pair(ops.dense(u_out), ops.dense(res_out), self.out_conv[0], head_conv)
            if z is not None:
                return z
            return self.out_conv(ops.cat_pad(u_out, res_out))
        if (isinstance(self.Upsample, (EUCB, EUCB2)) and not has_hooks(self.Upsample, self.up4, self.up4.conv)
                and res.shape[2] * 2 == h and res.shape[3] * 2 == w and self.Upsample.out_channels == c_res):
            # Feature Merge without copies: both branches write their result into its channel window of the merged
            # tensor (S/C:280 torch.cat([u_out, res_out], dim=1))
            cat = ops.empty_nhwc(n, c_u + c_res, h, w, x.dtype, x.device)
            wu, wr = ops.alias_window(cat, 0, c_u), ops.alias_window(cat, c_u, c_res)
            with on_side():
                res_out = self.Upsample(res, out=wr)
            u_out = unet_branch_forward(self, x, out=wu)
            if fork is not None:
                fork.join(res_out)
            if u_out.data_ptr() == wu.data_ptr() and res_out.data_ptr() == wr.data_ptr():
                return self.out_conv(ops.cat_windows(u_out, res_out, cat))
            return self.out_conv(ops.cat_pad(ops.dense(u_out), ops.dense(res_out)))
        with on_side():
            res_out = self.Upsample(res)
        u_out = unet_branch_forward(self, x)
        if fork is not None:
            fork.join(res_out)
        return self.out_conv(ops.cat_pad(u_out, res_out))


class UResNet(_TwoBranch):
    """S/C:223-284 (Upsample = EUCB, depthwise)."""

    def __init__(self, seg_classes=1, in_channels=3, base_c=32):
        super().__init__(seg_classes, in_channels, base_c)
        self.name = "U-ResNet"

    def _upsample(self):
        return EUCB(in_channels=32, out_channels=16, scale_factor=2)


class UResNetNewEUCB(_TwoBranch):
    """S/C:285-346 (Upsample = EUCB*, grouped)."""

    def __init__(self, seg_classes=1, in_channels=3, base_c=32):
        super().__init__(seg_classes, in_channels, base_c)
        self.name = "U-ResNetNewEUCB"

    def _upsample(self):
        return EUCB2(in_channels=32, out_channels=16, scale_factor=2)


class UResNetNoEUCB(_TwoBranch):
    """S/C:347-408 (ResLayer4 -> 16 channels, plain bilinear x2)."""
    _res4 = 16

    def __init__(self, seg_classes=1, in_channels=3, base_c=32):
        super().__init__(seg_classes, in_channels, base_c)
        self.name = "U-ResNet-NoEUCB"

    def _upsample(self):
        return hnn.Upsample(scale_factor=2, mode="bilinear")


class UNet(hnn.Module):
    """S/C:409-440 (the second `UNet` definition, which is the one importable from the reference file)."""
    _name = "U-Net"

    def __init__(self, seg_classes=1, in_channels=3, base_c=32):
        super().__init__()
        self.name = self._name
        self.compute_dtype = hnn.get_default_dtype()
        add_unet_branch(self, in_channels, base_c)
        _add_ups(self, base_c)
        self.out_conv = OutConv(base_c, seg_classes)

    set_compute_dtype = ResTrunk.set_compute_dtype

    def forward(self, x):
        x = ops.as_activation(x, self.compute_dtype)
        return self.out_conv(unet_branch_forward(self, x))


class UResNetNoResBlock(UNet):
    """S/I:409-440."""
    _name = "UResNetNoResBlock"
