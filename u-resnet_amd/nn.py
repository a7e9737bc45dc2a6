"""Leaf layers with torch.nn's constructors, parameters and state_dict keys, and HIP forwards.

Each class subclasses its torch.nn namesake so that parameter shapes, default initialisation
(hence `torch.manual_seed` reproducibility) and checkpoint keys are exactly the reference's;
only `forward` is replaced.  Configurations the reference's U-ResNet path never uses raise
NotImplementedError instead of silently falling back to ATen.
"""
import torch
import torch.nn as tnn

from . import _lib, ops

Sequential = tnn.Sequential
Module = tnn.Module

_default_dtype = torch.bfloat16


def set_default_dtype(dtype):
    """Activation dtype newly constructed top-level networks compute in (bf16 or fp32)."""
    global _default_dtype
    if dtype not in (torch.float32, torch.bfloat16):
        raise TypeError("compute dtype must be torch.float32 or torch.bfloat16")
    _default_dtype = dtype


def get_default_dtype():
    return _default_dtype


def act_dtype(x, preferred=None):
    """dtype a block computes in when called on x: the model's, else x's own if supported."""
    if preferred is not None:
        return preferred
    return x.dtype if x.dtype in (torch.float32, torch.bfloat16) else torch.float32


def _single(v):
    if isinstance(v, (tuple, list)):
        if len(set(v)) != 1:
            raise NotImplementedError("only square kernels / symmetric strides and paddings are implemented (got %r)"
                                      % (v,))
        return int(v[0])
    return int(v)


class Conv2d(tnn.Conv2d):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._packs = ops.PackedWeights()
        if self.padding_mode != 'zeros' or _single(self.dilation) != 1 or isinstance(self.padding, str):
            raise NotImplementedError("Conv2d: only zero padding (integer) and dilation 1 are implemented")
        self._k, self._s, self._p = _single(self.kernel_size), _single(self.stride), _single(self.padding)
        if self.groups != 1:
            cpg = self.in_channels // self.groups
            if not (self.in_channels == self.out_channels and cpg in (1, 4) and self._k == 3 and self._s == 1
                    and self._p == 1 and self.bias is None):
                raise NotImplementedError("grouped Conv2d: only 3x3/s1/p1, in==out channels, 1 or 4 channels per "
                                          "group, no bias (the EUCB / EUCB* layers) is implemented")

    def raw(self, x, want_stats=False, out=None, want_alias=False):
        """(y, stats) -- stats are the per-channel partial sums for a following training-mode BatchNorm.  `out`:
        optional channel window (ops.alias_window) for y, honoured by the dense convolutions only (callers compare
        data_ptr())."""
        x = ops.as_activation(x, act_dtype(x))
        if self.groups != 1:
            if want_alias:
                raise NotImplementedError("grouped Conv2d has no input alias")
            if want_stats:
                return ops.gconv3x3(x, self.weight, True)
            return ops.gconv3x3(x, self.weight), None
        return ops.conv2d(x, self.weight, self.bias, self._packs, self._s, self._p, want_stats, out, want_alias)

    def forward(self, x):
        return self.raw(x)[0]


class ConvTranspose2d(tnn.ConvTranspose2d):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._packs = ops.PackedWeights()
        if not (_single(self.kernel_size) == 2 and _single(self.stride) == 2 and _single(self.padding) == 0
                and _single(self.output_padding) == 0 and self.groups == 1 and _single(self.dilation) == 1):
            raise NotImplementedError("ConvTranspose2d: only kernel_size=2, stride=2 (the `Up.up` layer) is implemented")

    def forward(self, x):
        x = ops.as_activation(x, act_dtype(x))
        return ops.conv_transpose2x2(x, self.weight, self.bias, self._packs)


class BatchNorm2d(tnn.BatchNorm2d):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if not self.affine:
            raise NotImplementedError("BatchNorm2d(affine=False) is not implemented")

    def forward(self, x):
        return ops.bn_act(ops.as_activation(x, act_dtype(x)), self, act=_lib.ACT_NONE)


class ReLU(tnn.ReLU):
    """inplace is accepted for signature compatibility; the kernel writes a new tensor."""

    def forward(self, x):
        return ops.activation(x, _lib.ACT_RELU)


class ReLU6(tnn.ReLU6):
    def forward(self, x):
        return ops.activation(x, _lib.ACT_RELU6)


class LeakyReLU(tnn.LeakyReLU):
    def forward(self, x):
        return ops.activation(x, _lib.ACT_LEAKYRELU, self.negative_slope)


class GELU(tnn.GELU):
    def forward(self, x):
        if self.approximate != 'none':
            raise NotImplementedError("GELU(approximate='tanh') is not implemented")
        return ops.activation(x, _lib.ACT_GELU)


class PReLU(tnn.PReLU):
    """Learnable negative slope, one per channel or one for all (same parameter name and shape as nn.PReLU)."""

    def forward(self, x):
        return ops.prelu(ops.as_activation(x, act_dtype(x)) if x.dim() == 4 else x, self.weight)


class Hardswish(tnn.Hardswish):
    def forward(self, x):
        return ops.activation(x, _lib.ACT_HSWISH)


class MaxPool2d(tnn.MaxPool2d):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if not (_single(self.kernel_size) == 2 and _single(self.stride) == 2 and _single(self.padding) == 0
                and _single(self.dilation) == 1 and not self.ceil_mode and not self.return_indices):
            raise NotImplementedError("MaxPool2d: only kernel 2, stride 2 (the `Down` layer) is implemented")

    def forward(self, x):
        return ops.maxpool2x2(ops.as_activation(x, act_dtype(x), allow_window=True))


class Upsample(tnn.Upsample):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if self.mode != 'bilinear' or self.align_corners or self.size is not None:
            raise NotImplementedError("Upsample: only mode='bilinear', align_corners=False with a scale_factor is implemented")
        self._scale = _single(self.scale_factor)
        if self._scale != float(_single(self.scale_factor)) or self._scale < 1:
            raise NotImplementedError("Upsample: integer scale factors only")

    def forward(self, x):
        return ops.upsample_bilinear(ops.as_activation(x, act_dtype(x)), self._scale)


class Linear(tnn.Linear):
    """y = x W^T + b as a 1x1 convolution over a 1x1 map (the GEMM kernel of the convolutions)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._packs = ops.PackedWeights()

    def forward(self, x):
        if x.dim() != 2:
            raise NotImplementedError("Linear: 2-D input expected")
        b, f = x.shape
        if f != self.in_features:                  # torch's addmm message (the reference's mis-sized heads raise it)
            raise RuntimeError("mat1 and mat2 shapes cannot be multiplied (%dx%d and %dx%d)"
                               % (b, f, self.in_features, self.out_features))
        x4 = x.contiguous().view(b, f, 1, 1)
        w4 = self.weight.view(self.out_features, self.in_features, 1, 1)
        y, _ = ops.conv2d(x4, w4, self.bias, self._packs, 1, 0, False)
        return y.view(b, self.out_features)


def has_hooks(*mods):
    """True if any of the modules carries forward hooks (then containers run layer by layer)."""
    for m in mods:
        if m._forward_hooks or m._forward_pre_hooks:
            return True
    return False


fuse_eval_bn = True          # False: inference runs convolution and BatchNorm as separate kernels (A/B comparisons)


def conv_bn_act(x, conv, bn, act=_lib.ACT_RELU, residual=None, out=None, pool=False, out_stats=False):
    """conv -> BatchNorm (training statistics taken from the conv kernel's epilogue) -> (+residual) -> act.
    Inference (bn in eval mode under torch.no_grad()): BatchNorm is folded into the convolution's epilogue."""
    if (fuse_eval_bn and not bn.training and not torch.is_grad_enabled() and isinstance(conv, Conv2d) and conv.groups == 1
            and conv.bias is None and (conv._k, conv._s, conv._p) == (3, 1, 1)):
        xa = ops.as_activation(x, act_dtype(x), allow_window=True)
        z = ops.conv3x3_bn_eval(xa, conv.weight, conv._packs, bn, act, residual, out)
        if z is not None:
            return z
    y, stats = conv.raw(x, want_stats=bn.training)
    return ops.bn_act(y, bn, residual=residual, stats=stats, act=act, out=out, pool=pool, out_stats=out_stats)
