"""U-ResNet building blocks (ResBlock, UBlock halves, EUCB / EUCB*, heads) on the HIP operators.

Constructor signatures, sub-module names and state_dict keys follow the reference files
(S/C = Segmentation/Cityscapes/modules/U_ResNet.py, C/C = Classification/modules/U_ResNet_CIFAR.py).
Containers run a fused schedule (conv with BN statistics in its epilogue, then one
normalise+residual+ReLU pass); when a forward hook is registered on any child they run child by
child instead, so feature-map hooks see every intermediate tensor.
"""
import torch

from .. import _lib, ops
from .. import nn as hnn
from ..nn import act_dtype, conv_bn_act, has_hooks

_ACT_CODES = {hnn.ReLU: _lib.ACT_RELU}


def channel_shuffle(x, groups):
    """S/C:6-15 (identity, returned as is, when groups == C)."""
    return ops.channel_shuffle(ops.as_activation(x, act_dtype(x)), groups)


def act_layer(act, inplace=False, neg_slope=0.2, n_prelu=1):
    """S/C:17-34."""
    act = act.lower()
    if act == 'relu':
        return hnn.ReLU(inplace)
    if act == 'relu6':
        return hnn.ReLU6(inplace)
    if act == 'leakyrelu':
        return hnn.LeakyReLU(neg_slope, inplace)
    if act == 'gelu':
        return hnn.GELU()
    if act == 'hswish':
        return hnn.Hardswish(inplace)
    if act == 'prelu':
        return hnn.PReLU(num_parameters=n_prelu, init=neg_slope)
    raise NotImplementedError('activation layer [%s] is not found' % act)


def _cba(seq, x, i, out=None, pool=False, out_stats=False):
    """Run seq[i] (conv), seq[i+1] (BN), seq[i+2] (activation) fused when the activation is ReLU.  `out`: optional
    channel window for the result (honoured on the fused path only; callers compare data_ptr()).  pool: the caller
    max-pools the result next, out_stats: the caller normalises it with another BatchNorm next (ops.bn_act)."""
    conv, bn, act = seq[i], seq[i + 1], seq[i + 2]
    if type(act) in _ACT_CODES:
        return conv_bn_act(x, conv, bn, _ACT_CODES[type(act)], out=out, pool=pool, out_stats=out_stats)
    return act(conv_bn_act(x, conv, bn, _lib.ACT_NONE))


def _cba_conv1x1(seq, x, i, pw, out=None):
    """seq[i] (conv), seq[i+1] (BN), seq[i+2] (ReLU) followed by the 1x1 convolution `pw`: the BatchNorm+ReLU pass is folded
    into the 1x1 kernels' operand load where they cover the shape (ops.bn_relu_conv1x1), so the normalised tensor is not
    materialised; None when that does not apply (the caller then runs the modules one by one)."""
    conv, bn, act = seq[i], seq[i + 1], seq[i + 2]
    if (type(act) is not hnn.ReLU or not isinstance(pw, hnn.Conv2d) or pw.stride != (1, 1) or pw.padding != (0, 0)
            or pw.groups != 1 or not ops.fuse_bn_into_conv1x1):
        return None
    n, _, h, w = x.shape
    if n * h * w < 1 << 16 or x.dtype != torch.bfloat16:          # (below that the streaming backward-weight kernel is not used)
        return None
    y, stats = conv.raw(x, want_stats=bn.training)
    z = ops.bn_relu_conv1x1(y, bn, pw, stats=stats, out=out)
    if z is None:                                                # not covered after all: finish the stage the usual way
        a = ops.bn_act(y, bn, stats=stats, act=_lib.ACT_RELU)
        return pw.raw(a, out=out)[0] if out is not None else pw(a)
    return z


class EUCB(hnn.Module):
    """S/C:36-57: bilinear up -> depthwise 3x3 -> BN -> act -> shuffle(groups=C) -> 1x1 conv."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, activation='relu', scale_factor=2):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.up_dwc = hnn.Sequential(
            hnn.Upsample(scale_factor=scale_factor, mode="bilinear"),
            hnn.Conv2d(in_channels, in_channels, kernel_size=kernel_size, stride=stride, padding=kernel_size // 2,
                       groups=in_channels, bias=False),
            hnn.BatchNorm2d(in_channels),
            act_layer(activation, inplace=True))
        self.pwc = hnn.Sequential(hnn.Conv2d(in_channels, out_channels, kernel_size=1, stride=1, padding=0, bias=True))

    def forward(self, x, out=None):
        """out: optional channel window for the result (honoured when no hooks are registered)."""
        x = ops.as_activation(x, act_dtype(x))
        hooked = has_hooks(self.up_dwc, *self.up_dwc, self.pwc, *self.pwc)
        if hooked:
            x = self.up_dwc(x)
        else:
            u = self.up_dwc[0](x)
            # (the shuffle between the stage and the projection is the identity: groups == channels, S/C:55)
            z = _cba_conv1x1(self.up_dwc, u, 1, self.pwc[0], out)
            if z is not None:
                return z
            x = _cba(self.up_dwc, u, 1)
        x = channel_shuffle(x, self.in_channels)
        if out is not None and not hooked:
            return self.pwc[0].raw(x, out=out)[0]
        return self.pwc(x)


class EUCB2(hnn.Module):
    """EUCB*: grouped 3x3 convs (groups = C/4).  stages=2: S/C:59-87; stages=3: C/C:64-100."""
    stages = 2

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, activation='relu', scale_factor=2):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels

        def stage():
            return [hnn.Conv2d(in_channels, in_channels, kernel_size=kernel_size, stride=stride,
                               padding=kernel_size // 2, groups=in_channels // 4, bias=False),
                    hnn.BatchNorm2d(in_channels), act_layer(activation, inplace=True)]

        self.up_dwc1 = hnn.Sequential(hnn.Upsample(scale_factor=scale_factor, mode="bilinear"), *stage())
        self.up_dwc2 = hnn.Sequential(*stage())
        if self.stages == 3:
            self.up_dwc3 = hnn.Sequential(*stage())
        self.pwc = hnn.Sequential(hnn.Conv2d(in_channels, out_channels, kernel_size=1, stride=1, padding=0, bias=True))

    def _seqs(self):
        return [self.up_dwc1, self.up_dwc2] + ([self.up_dwc3] if self.stages == 3 else [])

    def forward(self, x, out=None):
        """out: optional channel window for the result (honoured when no hooks are registered)."""
        x = ops.as_activation(x, act_dtype(x))
        hooked = has_hooks(self.pwc, *self.pwc, *[m for s in self._seqs() for m in (s, *s)])
        seqs = self._seqs()
        for k, seq in enumerate(seqs):
            if k > 0:
                x = channel_shuffle(x, self.in_channels)
            if hooked:
                x = seq(x)
            elif k == 0:
                x = _cba(seq, seq[0](x), 1)
            elif k == len(seqs) - 1:
                # the last stage feeds the 1x1 projection directly (S/C:84-86)
                z = _cba_conv1x1(seq, x, 0, self.pwc[0], out)
                if z is not None:
                    return z
                x = _cba(seq, x, 0)
            else:
                x = _cba(seq, x, 0)
        if out is not None and not hooked:
            return self.pwc[0].raw(x, out=out)[0]
        return self.pwc(x)


class EUCB2x3(EUCB2):
    stages = 3


class DoubleConv(hnn.Sequential):
    """S/C:89-100: (conv3x3 -> BN -> ReLU) x 2."""

    def __init__(self, in_channels, out_channels, mid_channels=None):
        if mid_channels is None:
            mid_channels = out_channels
        super().__init__(
            hnn.Conv2d(in_channels, mid_channels, kernel_size=3, padding=1, bias=False),
            hnn.BatchNorm2d(mid_channels),
            hnn.ReLU(inplace=True),
            hnn.Conv2d(mid_channels, out_channels, kernel_size=3, padding=1, bias=False),
            hnn.BatchNorm2d(out_channels),
            hnn.ReLU(inplace=True))

    def forward(self, x, out=None, pool=False, out_stats=False):
        """out: optional channel window for the result; pool: the caller max-pools the result next, out_stats: the caller
        normalises it with another BatchNorm next (see _cba)."""
        x = ops.as_activation(x, act_dtype(x))
        if has_hooks(*self):
            return super().forward(x)
        return _cba(self, _cba(self, x, 0), 3, out, pool, out_stats)


class Down(hnn.Sequential):
    """S/C:102-107: MaxPool2d(2,2) -> DoubleConv."""

    def __init__(self, in_channels, out_channels):
        super().__init__(hnn.MaxPool2d(2, stride=2), DoubleConv(in_channels, out_channels))


class Up(hnn.Module):
    """S/C:109-126: ConvTranspose2d(2,2) -> pad to the skip's size -> cat([skip, up]) -> DoubleConv."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.up = hnn.ConvTranspose2d(in_channels, in_channels // 2, kernel_size=2, stride=2)
        self.conv = DoubleConv(in_channels, out_channels)

    def forward(self, x1, x2, cat=None, out=None, out_stats=False):
        """cat: the pre-allocated concatenated tensor whose first channel window already holds x2 (x2 is then
        an alias of that window, see unet_branch_forward).  out: optional channel window for the result.
        out_stats: the caller normalises the result with another BatchNorm next (see _cba)."""
        if not has_hooks(self.up) and x2.shape[2] == 2 * x1.shape[2] and x2.shape[3] == 2 * x1.shape[3]:
            # the transposed convolution writes straight into its half of the concatenated tensor
            x1 = ops.as_activation(x1, act_dtype(x1))
            if cat is None:
                x2 = ops.as_activation(x2, x1.dtype)
            return self.conv(ops.conv_transpose2x2_cat(x1, self.up.weight, self.up.bias, self.up._packs, x2, cat), out=out,
                             out_stats=out_stats)
        x1 = self.up(x1)
        x2 = ops.as_activation(x2, x1.dtype)
        return self.conv(ops.cat_pad(x2, x1), out=out, out_stats=out_stats)


class OutConv(hnn.Sequential):
    """Segmentation head S/C:127-133: BN -> ReLU -> 1x1 conv (+bias)."""

    def __init__(self, in_channels, num_classes):
        super().__init__(hnn.BatchNorm2d(in_channels), hnn.ReLU(), hnn.Conv2d(in_channels, num_classes, kernel_size=1))

    def forward(self, x):
        x = ops.as_activation(x, act_dtype(x))
        if has_hooks(*self):
            return super().forward(x)
        conv = self[2]
        if conv.stride == (1, 1) and conv.padding == (0, 0) and conv.groups == 1 and conv.dilation == (1, 1):
            # the normalise pass folded into the convolution's operand load (head of 80 channels at full resolution)
            z = ops.bn_relu_conv1x1(x, self[0], conv)
            if z is not None:
                return z
        return conv(ops.bn_act(x, self[0], act=_lib.ACT_RELU))


class OutConvPlain(hnn.Sequential):
    """Classification head C/C:148-152: a bare 1x1 conv (+bias)."""

    def __init__(self, in_channels, num_classes):
        super().__init__(hnn.Conv2d(in_channels, num_classes, kernel_size=1))


class BasicBlock(hnn.Module):
    """ResBlock S/C:135-160: relu(BN(conv(relu(BN(conv(x))))) + shortcut(x))."""

    def __init__(self, inchannel, outchannel, stride=1):
        super().__init__()
        self.name = "Basic-Block-ResNet"
        self.left = hnn.Sequential(
            hnn.Conv2d(inchannel, outchannel, kernel_size=3, stride=stride, padding=1, bias=False),
            hnn.BatchNorm2d(outchannel),
            hnn.ReLU(inplace=True),
            hnn.Conv2d(outchannel, outchannel, kernel_size=3, stride=1, padding=1, bias=False),
            hnn.BatchNorm2d(outchannel))
        self.shortcut = hnn.Sequential()
        if stride != 1 or inchannel != outchannel:
            self.shortcut = hnn.Sequential(
                hnn.Conv2d(inchannel, outchannel, kernel_size=1, stride=stride, bias=False),
                hnn.BatchNorm2d(outchannel))

    def forward(self, x):
        x = ops.as_activation(x, act_dtype(x))
        if has_hooks(self.left, *self.left, self.shortcut, *self.shortcut):
            out = ops.add(self.left(x), self.shortcut(x))
            return ops.activation(out, _lib.ACT_RELU)
        if fuse_residual_grad and x.requires_grad and x.is_cuda and self.left[0].groups == 1:
            # the shortcut reads an alias of x handed out by the first convolution, whose backward-data kernel then
            # adds the shortcut's gradient in its epilogue
            y, stats, xs = self.left[0].raw(x, want_stats=self.left[1].training, want_alias=True)
            out = ops.bn_act(y, self.left[1], stats=stats, act=_lib.ACT_RELU)
        else:
            out, xs = _cba(self.left, x, 0), x
        sc = xs if len(self.shortcut) == 0 else conv_bn_act(xs, self.shortcut[0], self.shortcut[1], _lib.ACT_NONE)
        # second conv's BN, the residual add and the final ReLU are one pass
        return conv_bn_act(out, self.left[3], self.left[4], _lib.ACT_RELU, residual=sc)


class Bottleneck(hnn.Module):
    """S/C:162-188 (defined by the reference, not used by any U-ResNet variant); runs layer by layer."""

    def __init__(self, inchannel, outchannel, stride=1):
        super().__init__()
        self.name = 'Bottleneck'
        mid = int(outchannel / 4)
        self.left = hnn.Sequential(
            hnn.Conv2d(inchannel, mid, kernel_size=1, stride=stride, padding=0, bias=False), hnn.BatchNorm2d(mid),
            hnn.ReLU(inplace=True),
            hnn.Conv2d(mid, mid, kernel_size=3, stride=1, padding=1, bias=False), hnn.BatchNorm2d(mid),
            hnn.ReLU(inplace=True),
            hnn.Conv2d(mid, outchannel, kernel_size=1, stride=1, padding=0, bias=False), hnn.BatchNorm2d(outchannel))
        self.shortcut = hnn.Sequential()
        if stride != 1 or inchannel != outchannel:
            self.shortcut = hnn.Sequential(
                hnn.Conv2d(inchannel, outchannel, kernel_size=1, stride=stride, bias=False),
                hnn.BatchNorm2d(outchannel))

    def forward(self, x):
        x = ops.as_activation(x, act_dtype(x))
        return ops.activation(ops.add(self.left(x), self.shortcut(x)), _lib.ACT_RELU)


def stem(in_channels, selected_upsampling=False):
    """conv1: [SU = bilinear x2 (C/C:302-307)] -> 7x7 s2 conv -> BN -> ReLU (S/C:229-233)."""
    layers = [hnn.Upsample(scale_factor=2, mode='bilinear')] if selected_upsampling else []
    layers += [hnn.Conv2d(in_channels, 64, kernel_size=7, stride=2, padding=3, bias=False), hnn.BatchNorm2d(64),
               hnn.ReLU()]
    return hnn.Sequential(*layers)


def run_stem(seq, x):
    if has_hooks(seq, *seq):
        return seq(x)
    if len(seq) == 4:
        x = seq[0](x)
        return _cba(seq, x, 1)
    return _cba(seq, x, 0)


class ResTrunk(hnn.Module):
    """make_layer bookkeeping (S/C:252-258) shared by the top-level networks."""

    def make_layer(self, block, channels, num_blocks, stride):
        strides = [stride] + [1] * (num_blocks - 1)
        layers = []
        for stride in strides:
            layers.append(block(self.inchannel, channels, stride))
            self.inchannel = channels
        return hnn.Sequential(*layers)

    def _build_res(self, block, widths, strides):
        self.inchannel = 64
        for i, (c, s) in enumerate(zip(widths, strides), 1):
            setattr(self, 'ResLayer%d' % i, self.make_layer(block, c, 2, s))

    def _res_forward(self, x):
        out = run_stem(self.conv1, x)
        for i in (1, 2, 3, 4):
            out = getattr(self, 'ResLayer%d' % i)(out)
        return out

    def set_compute_dtype(self, dtype):
        import torch
        if dtype not in (torch.float32, torch.bfloat16):
            raise TypeError("compute dtype must be torch.float32 or torch.bfloat16")
        self.compute_dtype = dtype
        return self


def add_unet_branch(m, in_channels, base_c, levels=4):
    m.in_conv = DoubleConv(in_channels, base_c)
    for i in range(levels):
        setattr(m, 'down%d' % (i + 1), Down(base_c << i, base_c << (i + 1)))


def _skip_into_cat(block, x, up, pooled=None, pool_next=False):
    """Run the encoder stage `block` (DoubleConv or Down) on x and return (skip, cat): when possible the stage's last
    BatchNorm+ReLU writes the skip tensor straight into the first channel window of the tensor that `up` will
    concatenate it into (cat, else None), which removes the copy of the skip half of torch.cat (S/C:119-124).
    pooled: the Down stage's max-pool result when the caller has already computed it (_pool_and_skip).
    pool_next: the caller max-pools the skip tensor next (the following Down stage)."""
    down = isinstance(block, Down)
    dc = block[1] if down else block
    fused = (isinstance(dc, DoubleConv) and isinstance(up, Up) and not has_hooks(block, up, up.up, *dc)
             and not (down and has_hooks(*block)) and x.is_cuda and x.dim() == 4)
    if fused:
        n, _, h, w = x.shape
        if down:
            h, w = h // 2, w // 2
        c_skip, c_up = dc[3].out_channels, up.up.out_channels
        # the level below halves (floor) and the transposed convolution doubles: sizes must be even to line up
        fused = h % 2 == 0 and w % 2 == 0 and h > 0 and w > 0 and up.conv[0].in_channels == c_skip + c_up
    if not fused:
        if pooled is not None:
            return block[1](pooled), None
        return block(x), None
    dtype = act_dtype(x)
    cat = ops.empty_nhwc(n, c_skip + c_up, h, w, dtype, x.device)
    win = ops.alias_window(cat, 0, c_skip)
    skip = dc(block[0](x) if pooled is None else pooled, out=win, pool=pool_next) if down else dc(x, out=win, pool=pool_next)
    if skip.data_ptr() != win.data_ptr():          # the block could not honour the window
        return skip, None
    return skip, cat


fuse_pool_skip = True       # False: separate max-pool and gradient add (A/B comparisons)
fuse_residual_grad = True   # False: the two gradients of a residual block's input are summed by a separate add


def _pools_with_skip(down):
    return fuse_pool_skip and isinstance(down, Down) and not has_hooks(down, down[0])


def _pool_and_skip(down, x):
    """(max-pooled x for the stage `down`, x for the skip connection).  Without hooks both come from one operator whose
    backward sums the two gradients of x inside the pooling kernel; otherwise (None, x)."""
    if (_pools_with_skip(down) and x.is_cuda and x.dim() == 4
            and x.dtype in (torch.float32, torch.bfloat16) and x.requires_grad and ops._channel_pitch(x)):
        return ops.maxpool2x2_skip(x)
    return None, x


def unet_branch_forward(m, x, out=None, out_stats=False):
    """out: optional channel window for the branch's result; out_stats: the caller normalises the result with another
    BatchNorm next (see Up.forward)."""
    x1, c1 = _skip_into_cat(m.in_conv, x, m.up4, pool_next=_pools_with_skip(m.down1))
    p1, x1 = _pool_and_skip(m.down1, x1)
    x2, c2 = _skip_into_cat(m.down1, x1, m.up3, p1, pool_next=_pools_with_skip(m.down2))
    p2, x2 = _pool_and_skip(m.down2, x2)
    x3, c3 = _skip_into_cat(m.down2, x2, m.up2, p2, pool_next=_pools_with_skip(m.down3))
    p3, x3 = _pool_and_skip(m.down3, x3)
    x4, c4 = _skip_into_cat(m.down3, x3, m.up1, p3, pool_next=_pools_with_skip(m.down4))
    p4, x4 = _pool_and_skip(m.down4, x4)
    x5 = m.down4(x4) if p4 is None else m.down4[1](p4)
    u = m.up1(x5, x4, c4)
    u = m.up2(u, x3, c3)
    u = m.up3(u, x2, c2)
    return m.up4(u, x1, c1, out=out, out_stats=out_stats)
