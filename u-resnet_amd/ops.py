"""Autograd operators over the C ABI of liburesnet_hip.so.

Every operator takes / returns torch tensors that are logically NCHW but physically NHWC
(torch.channels_last), in the model's activation dtype (bf16 or fp32).  torch supplies device
memory, streams and the autograd tape only: all arithmetic runs in the hand-written HIP
kernels.  Parameters and their gradients are fp32 in PyTorch's native layouts, so
state_dicts are interchangeable with the reference's.
"""
import ctypes as C
import os
import weakref

import torch
from torch.autograd import Function

from . import _lib
from ._lib import ConvDesc, call, query

CL = torch.channels_last
_DT = {torch.float32: _lib.UR_F32, torch.bfloat16: _lib.UR_BF16}

_weights_epoch = 0          # bumped whenever this package's optimisers rewrite parameters in place


def bump_weights_epoch():
    global _weights_epoch
    _weights_epoch += 1


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dt(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError("unsupported activation dtype %s (use torch.float32 or torch.bfloat16)" % t.dtype)


def _p(t):
    return None if t is None else t.data_ptr()


def _require_gpu(t):
    if not t.is_cuda:
        raise _lib.UResNetHipError("u-resnet_amd operators run on the GPU only (got a %s tensor); "
                                   "there is no CPU fallback" % t.device)


def empty_nhwc(n, c, h, w, dtype, device):
    return torch.empty((n, c, h, w), dtype=dtype, device=device, memory_format=CL)


def _is_nhwc_dense(x):
    n, c, h, w = x.shape
    return x.stride() == (h * w * c, 1, w * c, c) or x.is_contiguous(memory_format=CL)


def to_nhwc_raw(x, dtype):
    """NCHW-contiguous (any supported dtype) -> dense NHWC tensor of `dtype`, via the transpose kernel."""
    _require_gpu(x)
    n, c, h, w = x.shape
    if x.dtype == dtype and _is_nhwc_dense(x):
        return x
    if _is_nhwc_dense(x) and not x.is_contiguous():
        # already channels-last, only the dtype differs: cast the flat buffer (a 1 x numel "transpose")
        y = empty_nhwc(n, c, h, w, dtype, x.device)
        call("ur_nchw_to_nhwc", x.data_ptr(), _dt(x), y.data_ptr(), _DT[dtype], 1, 1, 1, x.numel(), _stream())
        return y
    if not x.is_contiguous():
        x = x.contiguous()
    y = empty_nhwc(n, c, h, w, dtype, x.device)
    call("ur_nchw_to_nhwc", x.data_ptr(), _dt(x), y.data_ptr(), _DT[dtype], n, c, h, w, _stream())
    return y


def to_nchw_raw(x, dtype):
    """dense NHWC -> NCHW-contiguous tensor of `dtype`."""
    n, c, h, w = x.shape
    y = torch.empty((n, c, h, w), dtype=dtype, device=x.device)
    call("ur_nhwc_to_nchw", x.data_ptr(), _dt(x), y.data_ptr(), _DT[dtype], n, c, h, w, _stream())
    return y


class _ToNHWC(Function):
    @staticmethod
    def forward(ctx, x, dtype):
        ctx.in_dtype = x.dtype
        ctx.in_cl = _is_nhwc_dense(x) and not x.is_contiguous()
        return to_nhwc_raw(x, dtype)

    @staticmethod
    def backward(ctx, g):
        g = _g(g)
        if ctx.in_cl:
            return to_nhwc_raw(g, ctx.in_dtype), None
        return to_nchw_raw(g, ctx.in_dtype), None


def as_activation(x, dtype, allow_window=False):
    """Model-boundary conversion: any 4-D input -> dense NHWC tensor of the compute dtype.  With allow_window a
    channel window of a wider NHWC tensor (alias_window) passes through for the kernels that take a pitch."""
    if x.dim() != 4:
        raise ValueError("expected a 4-D NCHW tensor, got shape %s" % (tuple(x.shape),))
    if x.dtype == dtype and x.is_cuda and _is_nhwc_dense(x):
        return x
    if x.dtype == dtype and x.is_cuda and _channel_pitch(x):
        return x if allow_window else _Dense.apply(x)
    return _ToNHWC.apply(x, dtype)


def alias_window(buf, c0, c):
    """Tensor over channels [c0, c0 + c) of the dense NHWC tensor `buf`, sharing its storage but not registered as a
    view of it (autograd never sees the aliasing; the kernels that fill the windows write through raw pointers)."""
    n, ct, h, w = buf.shape
    t = torch.empty(0, dtype=buf.dtype, device=buf.device)
    t.set_(buf.untyped_storage(), buf.storage_offset() + c0, (n, c, h, w), (h * w * ct, 1, w * ct, ct))
    return t


def _channel_pitch(x):
    """Channel pitch of an NHWC tensor whose pixels are `pitch` elements apart (a channel window of a wider
    dense NHWC tensor), or 0 if x is not of that form."""
    n, c, h, w = x.shape
    s = x.stride()
    ld = s[3]
    if s[1] == 1 and ld >= c and s[2] == w * ld and s[0] == h * w * ld:
        return ld
    return 0


def _g(g):
    """Incoming gradient -> dense NHWC tensor (autograd may hand over NCHW, expanded or channel-window tensors)."""
    if _is_nhwc_dense(g):
        return g
    ld = _channel_pitch(g)
    if ld and g.is_cuda and g.dtype in _DT:
        n, c, h, w = g.shape
        y = empty_nhwc(n, c, h, w, g.dtype, g.device)
        call("ur_copy_channels", g.data_ptr(), ld, 0, h, w, y.data_ptr(), c, 0, h, w, 0, 0, n, c, _dt(g), _stream())
        return y
    return to_nhwc_raw(g.contiguous(), g.dtype)


class _Dense(Function):
    """Channel window of a wider NHWC tensor -> dense NHWC copy."""

    @staticmethod
    def forward(ctx, x):
        return _g(x)

    @staticmethod
    def backward(ctx, g):
        return g


def dense(x):
    """x itself when it is a dense NHWC tensor, else a dense copy of the channel window."""
    return x if _is_nhwc_dense(x) else _Dense.apply(x)


class _ToNCHW(Function):
    """NHWC activation -> NCHW-contiguous tensor (used where the reference flattens in NCHW order)."""

    @staticmethod
    def forward(ctx, x, dtype):
        ctx.dtype_in = x.dtype
        return to_nchw_raw(x, dtype)

    @staticmethod
    def backward(ctx, g):
        return to_nhwc_raw(g.contiguous(), ctx.dtype_in), None


def to_nchw(x, dtype=None):
    return _ToNCHW.apply(x, dtype or x.dtype)


# ------------------------------------------------------------------------------ dense convolution
def _desc(x_shape, cout, kh, kw, stride, pad, dtype, oh=None, ow=None):
    n, cin, h, w = x_shape
    d = ConvDesc()
    d.dtype = _DT[dtype]
    d.n, d.h, d.w, d.cin = n, h, w, cin
    d.oh = (h + 2 * pad - kh) // stride + 1 if oh is None else oh
    d.ow = (w + 2 * pad - kw) // stride + 1 if ow is None else ow
    d.cout = cout
    d.kh, d.kw, d.stride, d.pad = kh, kw, stride, pad
    d.in_ld, d.out_ld = cin, cout
    return d


batch_repack = True         # False: one packing launch per layer and direction (A/B comparisons)
_tile_images = []          # (weakref(PackedWeights), mode, weakref(weight)) of every tile-kernel weight image made so far
_tile_table = [None, None]  # (rows, device table) of the last batched repack


def _repack_stale_tile_images(device):
    """After an optimiser step every cached tile-kernel weight image is stale at once: rebuild them all with one
    launch (in place, the images keep their storage) instead of one launch per layer and direction."""
    rows, hits, live, nblocks = [], [], [], 0
    for ref, mode, wref in _tile_images:
        pw, w = ref(), wref()
        if pw is None or w is None:
            continue
        live.append((ref, mode, wref))
        hit = pw._cache.get(mode)
        if hit is None or not w.is_cuda or w.device != device or w.dtype != torch.float32 or not w.is_contiguous():
            continue
        key, packed = hit
        new_key = (w.data_ptr(), w._version, _weights_epoch) + key[3:]
        if key == new_key or key[0] != new_key[0]:
            continue
        dg = int(mode.endswith("dgrad"))
        rows.append((w.data_ptr(), packed.data_ptr(), key[4], key[5], dg | (2 if mode.startswith("wide") else 0), nblocks))
        nblocks += query("ur_conv3x3_tile_pack_blocks", key[4], key[5], dg)
        hits.append((pw, mode, new_key, packed))
    _tile_images[:] = live
    if not rows:
        return
    if _tile_table[0] != rows:
        if torch.cuda.is_current_stream_capturing():
            return                     # a synchronous upload cannot be captured: the caller packs its own image instead
        _tile_table[0] = rows
        _tile_table[1] = torch.tensor(rows, dtype=torch.int64).to(device)
    call("ur_conv3x3_tile_pack_weights_multi", _tile_table[1].data_ptr(), len(rows), nblocks, _stream())
    for pw, mode, new_key, packed in hits:
        pw._cache[mode] = (new_key, packed)


# Two-branch networks launch their ResBlock branch on a second stream (URESNET_BRANCH_STREAMS=0: everything on one stream).
branch_streams = os.environ.get("URESNET_BRANCH_STREAMS", "1") != "0"
_side_streams = {}
stream_probe_log = []       # what _concurrent_stream measured (one line per candidate stream)


def _overlaps(a, b, spin_us=300):
    """True if an idle wave on stream a and one on stream b run side by side (one spin time for both, not two)."""
    dev = a.device
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    call("ur_spin", 1, a.cuda_stream)          # first launches on a stream pay for its set-up: not measured
    call("ur_spin", 1, b.cuda_stream)
    torch.cuda.synchronize(dev)
    e0.record(a)
    call("ur_spin", spin_us, a.cuda_stream)
    call("ur_spin", spin_us, b.cuda_stream)
    e1.record(a)
    e2.record(b)
    torch.cuda.synchronize(dev)
    stream_probe_log.append("%.3f / %.3f ms for two %d us idle waves" % (e0.elapsed_time(e1), e0.elapsed_time(e2), spin_us))
    return max(e0.elapsed_time(e1), e0.elapsed_time(e2)) < 1.5e-3 * spin_us


def _concurrent_stream(*others, tries=8):
    """A stream whose launches run beside those of every stream in `others`.  The runtime maps streams onto a few hardware
    queues (four per process by default) and two streams on one queue execute in order, so which stream is independent
    of another depends on what else the process has created (communicator streams, torch's pools).  Measured, not
    assumed.  Falls back to the last candidate (correct, merely not concurrent) when none overlaps."""
    dev = others[0].device
    cand = torch.cuda.Stream(dev)
    if torch.cuda.is_current_stream_capturing():
        return cand
    for _ in range(tries):
        if all(_overlaps(o, cand) for o in others):
            break
        cand = torch.cuda.Stream(dev)
    return cand


class BranchFork:
    """Launch one of two independent branches of a network on a second stream.

        fork = BranchFork(x)            # x: tensors made on the current stream that the side branch reads
        with fork.branch(): a = side_branch(x)
        b = main_branch(x)
        fork.join(a)                    # the current stream waits; tensors made on the side stream are handed over

    Backward needs nothing: autograd runs every node on the stream its forward ran on and orders the streams itself.
    Weight images that went stale with the last optimiser step are rebuilt before the fork, on the current stream: the
    rebuild is one launch for all layers, so whichever branch triggered it would race with the other."""

    def __init__(self, *shared):
        dev = shared[0].device
        self.main = torch.cuda.current_stream(dev)
        self.side = _side_streams.get((dev, self.main.cuda_stream))
        if self.side is None:
            self.side = _side_streams[(dev, self.main.cuda_stream)] = _concurrent_stream(self.main)
        if batch_repack:
            _repack_stale_tile_images(dev)
        self.side.wait_stream(self.main)
        for t in shared:                       # their memory must not be reused while side-stream launches are pending
            t.record_stream(self.side)

    def branch(self):
        return torch.cuda.stream(self.side)

    def join(self, *made_on_side):
        self.main.wait_stream(self.side)
        for t in made_on_side:
            t.record_stream(self.main)


def two_branches(x, side_fn, main_fn):
    """(side_fn(x), main_fn(x)) for two independent branches fed by the same activation: with `branch_streams` the first
    is launched on the second stream (see BranchFork) and joined before this returns."""
    if branch_streams and x.is_cuda:
        fork = BranchFork(x)
        with fork.branch():
            a = side_fn(x)
        b = main_fn(x)
        fork.join(*(t for t in (a if isinstance(a, (tuple, list)) else (a,)) if torch.is_tensor(t)))
        return a, b
    return side_fn(x), main_fn(x)


class PackedWeights:
    """Per-layer cache of the weight matrices the GEMM kernels consume (activation dtype, k-major)."""

    def __init__(self):
        self._cache = {}

    def get_tile(self, weight, d, dgrad, wide=False):
        """Fragment-order image for the LDS-tiled 3x3 kernel (wide: in the piece order of the 128-channel kernel)."""
        mode = ("wide_" if wide else "tile_") + ("dgrad" if dgrad else "fwd")
        kern = "wide" if wide else "tile"
        key = (weight.data_ptr(), weight._version, _weights_epoch, d.dtype, d.cin, d.cout)
        hit = self._cache.get(mode)
        if hit is not None and hit[0] == key:
            return hit[1]
        if batch_repack and hit is not None and hit[0][0] == key[0] and hit[0][3:] == key[3:]:
            _repack_stale_tile_images(weight.device)           # same parameter, new values
            hit = self._cache.get(mode)
            if hit[0] == key:
                return hit[1]
        packed = torch.empty(query("ur_conv3x3_%s_packed_elems" % kern, C.byref(d)), dtype=torch.bfloat16, device=weight.device)
        call("ur_conv3x3_%s_pack_weight" % kern, weight.data_ptr(), packed.data_ptr(), C.byref(d), int(dgrad), _stream())
        if hit is None:
            _tile_images.append((weakref.ref(self), mode, weakref.ref(weight)))
        self._cache[mode] = (key, packed)
        return packed

    def get(self, weight, d, mode):
        """k-major matrix for the generic kernels.  `d` may describe the layer with its channel counts rounded
        up to a multiple of 8 (see _pad8): the parameter is then zero-padded to that shape before packing."""
        key = (weight.data_ptr(), weight._version, _weights_epoch, d.dtype, d.cin, d.cout)
        hit = self._cache.get(mode)
        if hit is not None and hit[0] == key:
            return hit[1]
        src = weight
        if mode in (_lib.PACK_FWD, _lib.PACK_DGRAD) and tuple(weight.shape[:2]) != (d.cout, d.cin):
            src = weight.new_zeros((d.cout, d.cin) + tuple(weight.shape[2:]))
            src[:weight.shape[0], :weight.shape[1]].copy_(weight.detach())
        elems = query("ur_conv2d_packed_elems", C.byref(d), mode)
        packed = torch.empty(elems, dtype=torch.bfloat16 if d.dtype == _lib.UR_BF16 else torch.float32,
                             device=weight.device)
        call("ur_conv2d_pack_weight", src.data_ptr(), packed.data_ptr(), C.byref(d), mode, _stream())
        self._cache[mode] = (key, packed)
        return packed


# Optional live kernel timing: when `kernel_timings` is a list, every dense-conv and BatchNorm kernel launch is
# bracketed by two events recorded on the stream it is launched on, together with its algorithmic FLOPs
# (tags starting with "hbm:" carry algorithmic bytes instead).
kernel_timings = None
kernel_timing_tags = None       # optional set of tags: only those launches are bracketed (each event costs about a microsecond)
use_tile_kernel = True          # False forces the generic implicit-GEMM kernel (A/B comparisons, tests)
use_wide_kernel = os.environ.get("URESNET_WIDE", "1") != "0"   # False: the 128-channel 3x3 kernel is never chosen (A/B comparisons)
use_pw_kernel = True            # False: 1x1 convolutions go through the generic kernel too


class _Timed:
    __slots__ = ("tag", "flops", "e0", "shape")

    def __init__(self, tag, flops, d=None):
        self.tag, self.flops, self.e0 = tag, flops, None
        self.shape = None if d is None else (d.n, d.h, d.w, d.cin, d.cout, d.kh, d.stride)

    def __enter__(self):
        if kernel_timings is not None and (kernel_timing_tags is None or self.tag in kernel_timing_tags):
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record(torch.cuda.current_stream())
        return self

    def __exit__(self, *exc):
        if self.e0 is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record(torch.cuda.current_stream())
            kernel_timings.append((self.tag, self.flops, self.e0, e1, self.shape))
        return False


def _conv_flops(d, transposed=False):
    if transposed:
        return 2.0 * d.n * d.h * d.w * d.cin * d.cout * 4
    return 2.0 * d.n * d.oh * d.ow * d.cout * d.cin * d.kh * d.kw


def grad_buffer(param, shape=None):
    """Tensor a backward-weight kernel writes the gradient of `param` into.  A data-parallel wrapper may register
    `param._ur_grad_slot`, a callable returning a fresh view of the parameter's slice of a gradient bucket: the kernel
    then fills the bucket directly and autograd adopts the view as `param.grad` without a copy.  Only when no gradient
    is held yet (an accumulation `grad += dw` must not alias its own operand); the callable returns None once the slice
    has been handed out in this step (a parameter used by two backward nodes: the second kernel gets its own tensor)."""
    shape = tuple(param.shape) if shape is None else tuple(shape)
    slot = getattr(param, "_ur_grad_slot", None)
    if slot is not None and param.grad is None and shape == tuple(param.shape):
        t = slot()
        if t is not None and t.dtype == param.dtype and t.device == param.device and t.is_contiguous() and tuple(t.shape) == shape:
            return t
    return torch.empty(shape, dtype=param.dtype, device=param.device)


def _workspace(nbytes, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


def _colsum(t2d_ptr, dtype_code, rows, c, ld, device, param=None):
    """Column sums of a rows x c matrix; `param`: the bias these are the gradient of (see grad_buffer)."""
    out = torch.empty(c, dtype=torch.float32, device=device) if param is None else grad_buffer(param, (c,))
    ws = _workspace(query("ur_colsum_workspace", rows, c), device)
    call("ur_colsum", t2d_ptr, dtype_code, rows, c, ld, out.data_ptr(), ws.data_ptr(), ws.numel(), _stream())
    return out


def _pad8(c):
    return (c + 7) // 8 * 8


def _pad_channels(x, cpad):
    """(N, C, H, W) NHWC activation -> (N, cpad, H, W) with zero channels appended.  The generic kernels gather
    8 channels per 16/32-byte load only when the channel count is a multiple of 8; a 3-channel image or a
    19-class gradient would otherwise be gathered element by element."""
    n, c, h, w = x.shape
    es = x.element_size()
    if _is_nhwc_dense(x) and (cpad * es) % 16 == 0 and c * es <= 128 and x.data_ptr() % 16 == 0:
        xp = empty_nhwc(n, cpad, h, w, x.dtype, x.device)          # one pass: rows widened and zero filled together
        call("ur_pad_channels", x.data_ptr(), c, xp.data_ptr(), cpad, n * h * w, _dt(x), _stream())
        return xp
    xp = empty_nhwc(n, cpad, h, w, x.dtype, x.device).zero_()
    call("ur_copy_channels", x.data_ptr(), c, 0, h, w, xp.data_ptr(), cpad, 0, h, w, 0, 0, n, c, _dt(x), _stream())
    return xp


def _crop_channels(x, c):
    """Dense (N, c, H, W) NHWC copy of the first c channels."""
    n, cx, h, w = x.shape
    y = empty_nhwc(n, c, h, w, x.dtype, x.device)
    call("ur_copy_channels", x.data_ptr(), cx, 0, h, w, y.data_ptr(), c, 0, h, w, 0, 0, n, c, _dt(x), _stream())
    return y


class _Conv2d(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, packs, stride, pad, want_stats, out, want_alias):
        _require_gpu(x)
        x_in = x
        cout, cin, kh, kw = weight.shape
        if x.shape[1] != cin:
            raise RuntimeError("conv2d: input has %d channels, weight expects %d" % (x.shape[1], cin))
        d = _desc(x.shape, cout, kh, kw, stride, pad, x.dtype)
        if d.oh <= 0 or d.ow <= 0:
            raise RuntimeError("conv2d: kernel %dx%d does not fit input %dx%d" % (kh, kw, d.h, d.w))
        win_ld = 0
        if out is not None:                          # (window,): a channel window of a wider tensor (alias_window)
            win_ld = _channel_pitch(out[0])
            if tuple(out[0].shape) != (d.n, cout, d.oh, d.ow) or out[0].dtype != x.dtype or not win_ld:
                raise RuntimeError("conv2d: `out` must be a channel window with the output's shape and dtype")
        tile = bias is None and use_tile_kernel and query("ur_conv3x3_tile_supported", C.byref(d), 0)
        tile_direct = False                          # the tile kernel writes the window itself when pitch and address allow
        if tile and win_ld and win_ld % 8 == 0 and out[0].data_ptr() % 16 == 0:
            d.out_ld = win_ld
            tile_direct = bool(query("ur_conv3x3_tile_supported", C.byref(d), 0))
            if not tile_direct:
                d.out_ld = cout
        y = out[0] if (win_ld and (not tile or tile_direct)) else empty_nhwc(d.n, cout, d.oh, d.ow, x.dtype, x.device)
        stats = None
        if tile:
            kern = "wide" if use_wide_kernel and query("ur_conv3x3_wide_preferred", C.byref(d), 0) else "tile"
            wp = packs.get_tile(weight, d, False, kern == "wide")
            if want_stats:
                rows = query("ur_conv3x3_%s_stats_rows" % kern, C.byref(d))
                stats = torch.empty((rows, cout, 2), dtype=torch.float32, device=x.device)
            with _Timed(kern + "_fwd", _conv_flops(d), d):
                call("ur_conv3x3_%s_fwd" % kern, x.data_ptr(), wp.data_ptr(), y.data_ptr(), _p(stats), C.byref(d), _stream())
        else:
            if cin % 8:
                x = _pad_channels(x, _pad8(cin))
                d = _desc(x.shape, cout, kh, kw, stride, pad, x.dtype)
            if win_ld:
                d.out_ld = win_ld
            if (use_pw_kernel and not want_stats and weight.dtype == torch.float32 and weight.is_contiguous()
                    and query("ur_conv1x1_pw_supported", C.byref(d), 0)):
                # HBM-bound pointwise layer: streaming kernel, reads the fp32 parameter directly
                with _Timed("pw_fwd", _conv_flops(d), d):
                    call("ur_conv1x1_pw_fwd", x.data_ptr(), weight.data_ptr(), cout, cin, _p(bias), y.data_ptr(), C.byref(d),
                         _stream())
            else:
                wp = packs.get(weight, d, _lib.PACK_FWD)
                if want_stats:
                    rows = query("ur_conv2d_stats_rows", C.byref(d))
                    stats = torch.empty((rows, cout, 2), dtype=torch.float32, device=x.device)
                with _Timed("igemm_fwd", _conv_flops(d), d):
                    call("ur_conv2d_fwd", x.data_ptr(), wp.data_ptr(), _p(bias), y.data_ptr(), _p(stats), C.byref(d), _stream())
        if win_ld and tile and not tile_direct:      # dense result -> rows of the window, win_ld elements apart
            call("ur_copy_channels", y.data_ptr(), cout, 0, d.oh, d.ow, out[0].data_ptr(), win_ld, 0, d.oh, d.ow, 0, 0,
                 d.n, cout, _dt(y), _stream())
            y = out[0]
        ctx.save_for_backward(x, weight)          # x: as the kernel saw it (channel-padded on the generic path)
        ctx.d, ctx.packs, ctx.has_bias, ctx.bias = d, packs, bias is not None, bias
        ctx.bn_rec = getattr(x_in, "_ur_bn", None) if tile else None
        ctx.set_materialize_grads(False)          # no zero tensor for the (non-differentiable) statistics output
        # alias of the input for a second consumer (the shortcut of a residual block): its gradient comes back to this
        # node and is added in the backward-data epilogue instead of by a separate add
        alias = None
        if want_alias:
            alias = torch.empty(0, dtype=x_in.dtype, device=x_in.device)
            alias.set_(x_in.untyped_storage(), x_in.storage_offset(), x_in.size(), x_in.stride())
        if want_stats:
            ctx.mark_non_differentiable(stats)
        return y, stats if want_stats else None, alias

    @staticmethod
    def backward(ctx, dy, _dstats, dalias):
        if dy is None:
            return dalias, None, None, None, None, None, None, None, None
        x, weight = ctx.saved_tensors
        d = ctx.d
        cout, cin = weight.shape[0], weight.shape[1]
        d.out_ld = cout
        dx = dw = db = None
        tile_d = use_tile_kernel and query("ur_conv3x3_tile_supported", C.byref(d), 1)
        tile_w = use_tile_kernel and query("ur_wgrad3x3_tile_supported", C.byref(d))
        # the generic kernels read a channel window of a wider gradient in place (the backward of a torch.cat)
        dy_ld = 0
        if not (tile_d or tile_w) and cout % 8 == 0 and dy.is_cuda and dy.dtype == x.dtype and not _is_nhwc_dense(dy):
            dy_ld = _channel_pitch(dy)
        if dy_ld:
            d.out_ld = dy_ld
        else:
            dy = _g(dy)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = _colsum(dy.data_ptr(), d.dtype, d.n * d.oh * d.ow, cout, d.out_ld, x.device, ctx.bias)
        if cout % 8 and not (tile_d and tile_w):
            dy = _pad_channels(dy, _pad8(cout))
            d = _desc(x.shape, _pad8(cout), d.kh, d.kw, d.stride, d.pad, x.dtype)
        if ctx.needs_input_grad[0]:
            dx = empty_nhwc(d.n, d.cin, d.h, d.w, x.dtype, x.device)
            if tile_d:
                add_ld = 0
                if dalias is not None and dalias.dtype == x.dtype and dalias.data_ptr() % 16 == 0:
                    add_ld = _channel_pitch(dalias)
                add_ok = bool(add_ld and add_ld % 8 == 0)
                rec = ctx.bn_rec
                fuse = (fuse_bn_bwd_sums and rec is not None and not rec.window and (dalias is None or add_ok)
                        and tuple(rec.y.shape) == tuple(dx.shape) and rec.y.dtype == dx.dtype and _is_nhwc_dense(rec.y)
                        and (not rec.use_z or (_is_nhwc_dense(x) and x.data_ptr() % 16 == 0)))
                # the 128-channel kernel has the plain, the addend and the mask-from-y BatchNorm epilogues
                wide = (use_wide_kernel and not (fuse and (rec.use_z or dalias is not None))
                        and bool(query("ur_conv3x3_wide_preferred", C.byref(d), 1)))
                wp = ctx.packs.get_tile(weight, d, True, wide)
                with _Timed(("wide" if wide else "tile") + ("_dgrad_bn" if fuse else "_dgrad"), _conv_flops(d), d):
                    if wide and fuse:
                        rows = query("ur_conv3x3_wide_stats_rows", C.byref(d))
                        partial = torch.empty((rows, d.cin, 2), dtype=torch.float32, device=x.device)
                        call("ur_conv3x3_wide_dgrad_bnsums", dy.data_ptr(), wp.data_ptr(), dx.data_ptr(), rec.y.data_ptr(), d.cin,
                             rec.mean.data_ptr(), rec.invstd.data_ptr(), rec.gamma.data_ptr(), rec.beta.data_ptr(),
                             partial.data_ptr(), C.byref(d), _stream())
                        rec.partial, rec.dz_ptr, rec.dz_version = partial, dx.data_ptr(), dx._version
                    elif wide and add_ok:
                        call("ur_conv3x3_wide_dgrad_add", dy.data_ptr(), wp.data_ptr(), dx.data_ptr(), dalias.data_ptr(), add_ld,
                             C.byref(d), _stream())
                        dalias = None
                    elif wide:
                        call("ur_conv3x3_wide_dgrad", dy.data_ptr(), wp.data_ptr(), dx.data_ptr(), C.byref(d), _stream())
                    elif fuse:
                        # dx is the gradient of a BatchNorm+ReLU output: its backward sums come out of this kernel's epilogue
                        rows = query("ur_conv3x3_tile_dgrad_stats_rows", C.byref(d))
                        partial = torch.empty((rows, d.cin, 2), dtype=torch.float32, device=x.device)
                        call("ur_conv3x3_tile_dgrad_bnsums", dy.data_ptr(), wp.data_ptr(), dx.data_ptr(),
                             dalias.data_ptr() if dalias is not None else None, add_ld if dalias is not None else 0,
                             rec.y.data_ptr(), d.cin, x.data_ptr() if rec.use_z else None, rec.mean.data_ptr(),
                             rec.invstd.data_ptr(),
                             rec.gamma.data_ptr(), rec.beta.data_ptr(), partial.data_ptr(), C.byref(d), _stream())
                        rec.partial, rec.dz_ptr, rec.dz_version = partial, dx.data_ptr(), dx._version
                        dalias = None
                    elif add_ok:
                        call("ur_conv3x3_tile_dgrad_add", dy.data_ptr(), wp.data_ptr(), dx.data_ptr(), dalias.data_ptr(), add_ld,
                             C.byref(d), _stream())
                        dalias = None
                    else:
                        call("ur_conv3x3_tile_dgrad", dy.data_ptr(), wp.data_ptr(), dx.data_ptr(), C.byref(d), _stream())
            elif (use_pw_kernel and weight.dtype == torch.float32 and weight.is_contiguous()
                    and query("ur_conv1x1_pw_supported", C.byref(d), 1)):
                with _Timed("pw_dgrad", _conv_flops(d), d):
                    call("ur_conv1x1_pw_dgrad", dy.data_ptr(), weight.data_ptr(), cout, cin, dx.data_ptr(), C.byref(d), _stream())
            else:
                wp = ctx.packs.get(weight, d, _lib.PACK_DGRAD)
                with _Timed("igemm_dgrad", _conv_flops(d), d):
                    call("ur_conv2d_dgrad", dy.data_ptr(), wp.data_ptr(), dx.data_ptr(), C.byref(d), _stream())
            if d.cin != cin:
                dx = _crop_channels(dx, cin)
            if dalias is not None:
                dalias = _g(dalias)
                call("ur_add_inplace", dx.data_ptr(), dalias.data_ptr(), _dt(dx), dx.numel(), _stream())
        if ctx.needs_input_grad[1]:
            dw = grad_buffer(weight, (d.cout, d.cin, d.kh, d.kw))
            if tile_w:
                ws = _workspace(query("ur_wgrad3x3_tile_workspace", C.byref(d)), x.device)
                # (supported == 2: the 32 x 32-block kernel for layers with 32 channels on a side)
                if tile_w == 1 and kernel_timings is not None and (kernel_timing_tags is None or "tile_wgrad" in kernel_timing_tags):
                    # timed: the library records `mid` between its matrix-core kernel and the slab reduction, so each of
                    # the two gets its own duration (the FLOPs belong to the first, the second is a pass over the slabs)
                    st = torch.cuda.current_stream()
                    e0, mid, e1 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                    mid.record(st)                  # creates the handle; recorded again by the library
                    e0.record(st)
                    call("ur_wgrad3x3_tile_marked", x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), ws.numel(),
                         C.byref(d), C.c_void_p(mid.cuda_event), _stream())
                    e1.record(st)
                    shape = (d.n, d.h, d.w, d.cin, d.cout, d.kh, d.stride)
                    kernel_timings.append(("tile_wgrad", _conv_flops(d), e0, mid, shape))
                    kernel_timings.append(("hbm:wgrad3x3_reduce", float(query("ur_wgrad3x3_tile_workspace", C.byref(d)) + dw.numel() * 4),
                                           mid, e1, None))
                else:
                    with _Timed("tile_wgrad" if tile_w == 1 else "tile_wgrad32", _conv_flops(d), d):
                        call("ur_wgrad3x3_tile", x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), ws.numel(),
                             C.byref(d), _stream())
            elif use_pw_kernel and query("ur_conv1x1_pw_wgrad_supported", C.byref(d)):
                ws = _workspace(query("ur_conv1x1_pw_wgrad_workspace", C.byref(d)), x.device)
                with _Timed("pw_wgrad", _conv_flops(d), d):
                    call("ur_conv1x1_pw_wgrad", x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), ws.numel(),
                         C.byref(d), _stream())
            else:
                ws = _workspace(query("ur_conv2d_wgrad_workspace", C.byref(d)), x.device)
                with _Timed("wgrad", _conv_flops(d), d):
                    call("ur_conv2d_wgrad", x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), ws.numel(),
                         C.byref(d), _stream())
            if (d.cout, d.cin) != (cout, cin):
                dw = dw[:cout, :cin].contiguous()
        return dx, dw, db, None, None, None, None, None, None


def conv2d(x, weight, bias, packs, stride=1, pad=0, want_stats=False, out=None, want_alias=False):
    """(y, stats or None[, alias]).  `out`: optional channel window (alias_window) that receives the result.
    want_alias: also return an alias of x for a second consumer, whose gradient is then summed in this convolution's
    backward-data kernel."""
    y, stats, alias = _Conv2d.apply(x, weight, bias, packs, stride, pad, want_stats, None if out is None else (out,),
                                    want_alias)
    return (y, stats, alias) if want_alias else (y, stats)


def conv3x3_bn_eval(x, weight, packs, bn, act, residual=None, out=None):
    """Inference only (no autograd): act(BN_eval(conv3x3(x)) (+ residual)) in ONE kernel -- the fixed statistics are
    folded to a per-channel multiply-add in the tile kernel's epilogue, so neither the raw convolution output nor a
    BatchNorm pass touches memory.  Returns None when the layer is not one the tile kernel takes (the caller then runs
    convolution and BatchNorm separately)."""
    if (not use_tile_kernel or torch.is_grad_enabled() or x.dtype != torch.bfloat16 or not x.is_cuda or x.dim() != 4
            or act not in (_lib.ACT_NONE, _lib.ACT_RELU) or bn.running_mean is None or tuple(weight.shape[2:]) != (3, 3)):
        return None
    cout, cin = weight.shape[0], weight.shape[1]
    n, cx, h, w = x.shape
    in_ld = _channel_pitch(x)
    if cx != cin or not in_ld:
        return None
    d = _desc(x.shape, cout, 3, 3, 1, 1, x.dtype)
    d.in_ld = in_ld
    res_ld = 0
    if residual is not None:
        res_ld = _channel_pitch(residual) if (residual.dtype == x.dtype and residual.is_cuda) else 0
        if tuple(residual.shape) != (n, cout, h, w) or not res_ld or res_ld % 8 or residual.data_ptr() % 16:
            return None
    if out is None:
        z = empty_nhwc(n, cout, h, w, x.dtype, x.device)
    else:
        z = out
        d.out_ld = _channel_pitch(z)
        if tuple(z.shape) != (n, cout, h, w) or z.dtype != x.dtype or not d.out_ld:
            raise RuntimeError("conv3x3_bn_eval: `out` must be a channel window with the output's shape and dtype")
    if not query("ur_conv3x3_tile_supported", C.byref(d), 0):
        return None
    scale = torch.empty(cout, dtype=torch.float32, device=x.device)
    shift = torch.empty(cout, dtype=torch.float32, device=x.device)
    call("ur_bn_eval_affine", bn.running_mean.data_ptr(), bn.running_var.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(),
         cout, bn.eps, scale.data_ptr(), shift.data_ptr(), _stream())
    wp = packs.get_tile(weight, d, False)
    with _Timed("tile_fwd", _conv_flops(d), d):
        call("ur_conv3x3_tile_fwd_affine", x.data_ptr(), wp.data_ptr(), z.data_ptr(), scale.data_ptr(), shift.data_ptr(),
             int(act == _lib.ACT_RELU), _p(residual), res_ld, C.byref(d), _stream())
    return z


class _ConvT2x2(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, packs):
        _require_gpu(x)
        cin, cout, kh, kw = weight.shape
        if (kh, kw) != (2, 2) or x.shape[1] != cin:
            raise RuntimeError("conv_transpose2d: only kernel 2x2 stride 2 is implemented (weight %s, input %s)"
                               % (tuple(weight.shape), tuple(x.shape)))
        n, _, h, w = x.shape
        d = _desc(x.shape, cout, 2, 2, 2, 0, x.dtype, oh=2 * h, ow=2 * w)
        wp = packs.get(weight, d, _lib.PACK_CONVT_FWD)
        y = empty_nhwc(n, cout, 2 * h, 2 * w, x.dtype, x.device)
        with _Timed("igemm_fwd", _conv_flops(d, True), d):
            call("ur_convT2x2_fwd", x.data_ptr(), wp.data_ptr(), _p(bias), y.data_ptr(), C.byref(d), _stream())
        ctx.save_for_backward(x, weight)
        ctx.d, ctx.packs, ctx.has_bias, ctx.bias = d, packs, bias is not None, bias
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        d = ctx.d
        dy = _g(dy)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            wp = ctx.packs.get(weight, d, _lib.PACK_CONVT_DGRAD)
            dx = empty_nhwc(d.n, d.cin, d.h, d.w, x.dtype, x.device)
            with _Timed("igemm_dgrad", _conv_flops(d, True), d):
                call("ur_convT2x2_dgrad", dy.data_ptr(), wp.data_ptr(), dx.data_ptr(), C.byref(d), _stream())
        if ctx.needs_input_grad[1]:
            dw = grad_buffer(weight)
            ws = _workspace(query("ur_convT2x2_wgrad_workspace", C.byref(d)), x.device)
            with _Timed("wgrad", _conv_flops(d, True), d):
                call("ur_convT2x2_wgrad", x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), ws.numel(),
                     C.byref(d), _stream())
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = _colsum(dy.data_ptr(), d.dtype, d.n * d.oh * d.ow, d.cout, d.cout, x.device, ctx.bias)
        return dx, dw, db, None


def conv_transpose2x2(x, weight, bias, packs):
    return _ConvT2x2.apply(x, weight, bias, packs)


class _ConvT2x2Cat(Function):
    """torch.cat([skip, conv_transpose2x2(x)], dim=1) for equal spatial sizes (S/C:119-124 without padding).

    The transposed convolution scatters straight into the channel window [cs, cs+cout) of the concatenated
    tensor (the kernels take a channel pitch), and its backward kernels read the matching window of the incoming
    gradient in place: one copy per direction (the skip half) instead of two.
    """

    @staticmethod
    def forward(ctx, x, weight, bias, packs, skip, buf):
        _require_gpu(x)
        cin, cout, kh, kw = weight.shape
        n, _, h, w = x.shape
        cs = skip.shape[1]
        d = _desc(x.shape, cout, 2, 2, 2, 0, x.dtype, oh=2 * h, ow=2 * w)
        d.out_ld = cs + cout
        wp = packs.get(weight, d, _lib.PACK_CONVT_FWD)
        # buf = (tensor,): the concatenated tensor, whose window [0, cs) the producer of `skip` has already written
        out = empty_nhwc(n, cs + cout, 2 * h, 2 * w, x.dtype, x.device) if buf is None else buf[0]
        with _Timed("igemm_fwd", _conv_flops(d, True), d):
            call("ur_convT2x2_fwd", x.data_ptr(), wp.data_ptr(), _p(bias), out.data_ptr() + cs * out.element_size(),
                 C.byref(d), _stream())
        if buf is None:
            _copy_channels(skip, 0, cs, out, 0)
        ctx.save_for_backward(x, weight)
        ctx.d, ctx.packs, ctx.has_bias, ctx.cs, ctx.bias = d, packs, bias is not None, cs, bias
        return out

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        d, cs = ctx.d, ctx.cs
        g = _g(g)
        dy_ptr = g.data_ptr() + cs * g.element_size()
        dx = dw = db = dskip = None
        if ctx.needs_input_grad[0]:
            wp = ctx.packs.get(weight, d, _lib.PACK_CONVT_DGRAD)
            dx = empty_nhwc(d.n, d.cin, d.h, d.w, x.dtype, x.device)
            with _Timed("igemm_dgrad", _conv_flops(d, True), d):
                call("ur_convT2x2_dgrad", dy_ptr, wp.data_ptr(), dx.data_ptr(), C.byref(d), _stream())
        if ctx.needs_input_grad[1]:
            dw = grad_buffer(weight)
            ws = _workspace(query("ur_convT2x2_wgrad_workspace", C.byref(d)), x.device)
            with _Timed("wgrad", _conv_flops(d, True), d):
                call("ur_convT2x2_wgrad", x.data_ptr(), dy_ptr, dw.data_ptr(), ws.data_ptr(), ws.numel(), C.byref(d), _stream())
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = _colsum(dy_ptr, d.dtype, d.n * d.oh * d.ow, d.cout, d.out_ld, x.device, ctx.bias)
        if ctx.needs_input_grad[4]:
            # a channel window of g, not a copy: the skip tensor has a second consumer, so autograd adds this to
            # the other gradient in one pass; a lone consumer densifies it through _g()
            dskip = g[:, :cs]
        return dx, dw, db, None, dskip, None


def conv_transpose2x2_cat(x, weight, bias, packs, skip, buf=None):
    """cat([skip, conv_transpose2x2(x)], dim=1); skip must have exactly twice x's height and width.  With `buf` (the
    dense concatenated tensor whose first window already holds `skip`, see alias_window) nothing is copied."""
    return _ConvT2x2Cat.apply(x, weight, bias, packs, skip, None if buf is None else (buf,))


class _GConv3x3(Function):
    """Grouped 3x3, stride 1, pad 1, with 1 or 4 channels per group."""

    @staticmethod
    def forward(ctx, x, weight, want_stats):
        _require_gpu(x)
        c, cpg, kh, kw = weight.shape
        n, cx, h, w = x.shape
        if (kh, kw) != (3, 3) or cx != c or cpg not in (1, 4):
            raise RuntimeError("grouped conv: only 3x3 with 1 or 4 channels per group is implemented "
                               "(weight %s, input %s)" % (tuple(weight.shape), tuple(x.shape)))
        y = torch.empty_like(x, memory_format=CL)
        rows = query("ur_gconv3x3_stats_rows", _dt(x), n, h, w, c, cpg) if want_stats else 0
        stats = None
        if rows:          # BatchNorm partial sums from the kernel's epilogue instead of a separate pass over y
            stats = torch.empty((rows, c, 2), dtype=torch.float32, device=x.device)
            with _Timed("hbm:gconv3x3_fwd", 2 * x.numel() * x.element_size()):
                call("ur_gconv3x3_fwd_stats", x.data_ptr(), weight.data_ptr(), y.data_ptr(), stats.data_ptr(), _dt(x), n, h, w,
                     c, cpg, _stream())
            ctx.mark_non_differentiable(stats)
        else:
            with _Timed("hbm:gconv3x3_fwd", 2 * x.numel() * x.element_size()):
                call("ur_gconv3x3_fwd", x.data_ptr(), weight.data_ptr(), y.data_ptr(), _dt(x), n, h, w, c, cpg, _stream())
        ctx.save_for_backward(x, weight)
        ctx.set_materialize_grads(False)
        return y, stats

    @staticmethod
    def backward(ctx, dy, _dstats=None):
        if dy is None:
            return None, None, None
        x, weight = ctx.saved_tensors
        c, cpg = weight.shape[0], weight.shape[1]
        n, _, h, w = x.shape
        dy = _g(dy)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x, memory_format=CL)
            with _Timed("hbm:gconv3x3_dgrad", 2 * x.numel() * x.element_size()):
                call("ur_gconv3x3_dgrad", dy.data_ptr(), weight.data_ptr(), dx.data_ptr(), _dt(x), n, h, w, c, cpg,
                     _stream())
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(weight)
            ws = _workspace(query("ur_gconv3x3_wgrad_workspace", n, h, w, c, cpg), x.device)
            with _Timed("hbm:gconv3x3_wgrad", 2 * x.numel() * x.element_size()):
                call("ur_gconv3x3_wgrad", x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), ws.numel(), _dt(x),
                     n, h, w, c, cpg, _stream())
        return dx, dw, None


def gconv3x3(x, weight, want_stats=False):
    """y, or (y, stats or None) with want_stats: per-channel partial sums for a following training-mode BatchNorm."""
    y, stats = _GConv3x3.apply(x, weight, want_stats)
    return (y, stats) if want_stats else y


# ------------------------------------------------------------------------------ batch norm (+act, +residual)
fuse_bn_bwd_sums = True     # False: BatchNorm backward always runs its own reduction pass (A/B comparisons)
fuse_pool_bn_sums = os.environ.get("URESNET_POOL_BN_SUMS", "1") != "0"   # the max-pool backward kernel emits them too


class _BNRecord:
    """What the consumer of a BatchNorm+ReLU output needs to start that BatchNorm's backward pass: a 3x3 convolution whose
    backward-data kernel produces the gradient dz of the output can emit the per-channel sums (sum g, sum g * x_hat) from
    its epilogue (ur_conv3x3_tile_dgrad_bnsums) instead of a separate pass over dz and y.  The record travels as the
    attribute `_ur_bn` of the output tensor; `partial` / `dz_ptr` / `dz_version` are filled in by the convolution's
    backward and consumed by _BNAct.backward, which checks that the gradient it receives is that very tensor, unmodified
    (a second consumer of z would have made autograd add to it or replace it).  The record does not reference the output
    tensor itself (the tensor references the record): `use_z` tells the consumer that the ReLU mask must come from the
    stored output, which the consumer holds anyway as its saved input."""
    __slots__ = ("y", "use_z", "gamma", "beta", "mean", "invstd", "partial", "dz_ptr", "dz_version", "window")

    def __init__(self):
        self.partial = None
        self.window = False


class _BNAct(Function):
    @staticmethod
    def forward(ctx, y, gamma, beta, residual, stats, running_mean, running_var, nbt, training, momentum, eps, act, out, rec,
                pool=None, ostats=None):
        _require_gpu(y)
        n, c, h, w = y.shape
        rows = n * h * w
        dev = y.device
        mean = torch.empty(c, dtype=torch.float32, device=dev)
        invstd = torch.empty(c, dtype=torch.float32, device=dev)
        if training:
            if stats is None:
                nparts = query("ur_bn_stats_rows", rows, c)
                stats = torch.empty((nparts, c, 2), dtype=torch.float32, device=dev)
                call("ur_bn_stats", y.data_ptr(), _dt(y), rows, c, stats.data_ptr(), _stream())
            call("ur_bn_finalize", stats.data_ptr(), stats.shape[0], rows, c, eps, momentum, mean.data_ptr(),
                 invstd.data_ptr(), _p(running_mean), _p(running_var), _p(nbt), _stream())
        else:
            call("ur_bn_eval_prepare", running_mean.data_ptr(), running_var.data_ptr(), c, eps, mean.data_ptr(),
                 invstd.data_ptr(), _stream())
        # backward re-reads the activated tensor (densely) when a residual was added before the activation: that
        # combination writes a dense tensor first and copies it into the window
        via_dense = out is not None and act and residual is not None
        if out is None or via_dense:
            z, z_ld = torch.empty_like(y, memory_format=CL), c
        else:                                        # (window,): a channel window of a wider tensor (alias_window)
            z = out[0]
            z_ld = _channel_pitch(z)
            if tuple(z.shape) != tuple(y.shape) or z.dtype != y.dtype or not z_ld:
                raise RuntimeError("bn_act: `out` must be a channel window with y's shape and dtype")
        tb = y.numel() * y.element_size()            # bytes of one pass over the tensor
        if (pool is not None and residual is None and z_ld % 8 == 0 and z.data_ptr() % 16 == 0 and y.data_ptr() % 16 == 0
                and y.is_contiguous(memory_format=CL) and query("ur_bn_act_fwd_pool_supported", _dt(y), n, h, w, c)):
            # the caller max-pools the result next (maxpool2x2_skip): both come out of one pass
            pooled = empty_nhwc(n, c, h // 2, w // 2, y.dtype, dev)
            idx = torch.empty((n, h // 2, w // 2, c), dtype=torch.uint8, device=dev)
            with _Timed("hbm:bn_fwd_pool", tb * 2 + pooled.numel() * (pooled.element_size() + 1)):
                call("ur_bn_act_fwd_pool", y.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), act,
                     z.data_ptr(), z_ld, pooled.data_ptr(), idx.data_ptr(), _dt(y), n, h, w, c, _stream())
            pool.append((pooled, idx))
        elif (ostats is not None and residual is None and z_ld == c and z.data_ptr() % 16 == 0 and y.data_ptr() % 16 == 0
                and y.is_contiguous(memory_format=CL) and query("ur_bn_act_fwd_stats_supported", _dt(y), rows, c)):
            # a second BatchNorm normalises the result next (bn_relu_conv1x1_pair): its sums come out of this pass
            part = torch.empty((query("ur_bn_act_fwd_stats_rows", rows, c), c, 2), dtype=torch.float32, device=dev)
            with _Timed("hbm:bn_fwd_stats", tb * 2):
                call("ur_bn_act_fwd_stats", y.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), act,
                     z.data_ptr(), _dt(y), rows, c, part.data_ptr(), _stream())
            ostats.append(part)
        else:
            with _Timed("hbm:bn_fwd", tb * (3 if residual is not None else 2)):
                call("ur_bn_act_fwd", y.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                     _p(residual), act, z.data_ptr(), _dt(y), rows, c, z_ld, _stream())
        # the ReLU mask is recomputed from y in backward unless a residual was added before the activation
        ctx.save_for_backward(y, z if (act and residual is not None) else None, gamma, beta, mean, invstd)
        ctx.act, ctx.eval, ctx.has_res = act, not training, residual is not None
        ctx.rec = None
        if (rec is not None and training and act == _lib.ACT_RELU and y.dtype == torch.bfloat16 and c % 8 == 0
                and y.data_ptr() % 16 == 0):
            rec.y, rec.use_z = y, residual is not None
            rec.gamma, rec.beta, rec.mean, rec.invstd = gamma, beta, mean, invstd
            rec.window = not (out is None or via_dense)       # the result lives in a channel window of a wider tensor
            ctx.rec = rec
        if via_dense:
            if tuple(out[0].shape) != tuple(y.shape) or out[0].dtype != y.dtype or not _channel_pitch(out[0]):
                raise RuntimeError("bn_act: `out` must be a channel window with y's shape and dtype")
            n_, _, h_, w_ = y.shape
            call("ur_copy_channels", z.data_ptr(), c, 0, h_, w_, out[0].data_ptr(), _channel_pitch(out[0]), 0, h_, w_, 0, 0,
                 n_, c, _dt(z), _stream())
            return out[0]
        return z

    @staticmethod
    def backward(ctx, dz):
        y, z, gamma, beta, mean, invstd = ctx.saved_tensors
        n, c, h, w = y.shape
        rows = n * h * w
        dev = y.device
        # a channel window of a wider gradient (the backward of a torch.cat) is read in place
        dz_ld = _channel_pitch(dz) if (dz.is_cuda and dz.dtype == y.dtype) else 0
        if not dz_ld:
            dz, dz_ld = _g(dz), c
        tb = y.numel() * y.element_size()
        rec = ctx.rec
        if (rec is not None and rec.partial is not None and dz_ld == c and dz.data_ptr() == rec.dz_ptr
                and dz._version == rec.dz_version):
            # the backward-data kernel that produced dz has already summed g and g * x_hat per tile
            partial, rec.partial = rec.partial, None
            nparts = partial.shape[0]
        else:
            nparts = query("ur_bn_stats_rows", rows, c)
            partial = torch.empty((nparts, c, 2), dtype=torch.float32, device=dev)
            with _Timed("hbm:bn_bwd_reduce", tb * (3 if z is not None else 2)):
                call("ur_bn_act_bwd_reduce", dz.data_ptr(), y.data_ptr(), _p(z), mean.data_ptr(), invstd.data_ptr(),
                     gamma.data_ptr(), beta.data_ptr(), ctx.act, _dt(y), rows, c, dz_ld, partial.data_ptr(), _stream())
        if rec is not None:
            rec.partial = None
        sums = torch.empty((c, 2), dtype=torch.float32, device=dev)
        dgamma, dbeta = grad_buffer(gamma), grad_buffer(beta)
        call("ur_bn_bwd_finalize", partial.data_ptr(), nparts, c, sums.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
             _stream())
        dy = torch.empty_like(y, memory_format=CL)
        dres = torch.empty_like(y, memory_format=CL) if (ctx.has_res and ctx.needs_input_grad[3]) else None
        with _Timed("hbm:bn_bwd_apply", tb * (3 + (z is not None) + (dres is not None))):
            call("ur_bn_act_bwd_apply", dz.data_ptr(), y.data_ptr(), _p(z), mean.data_ptr(), invstd.data_ptr(),
                 gamma.data_ptr(), beta.data_ptr(), sums.data_ptr(), ctx.act, int(ctx.eval), dy.data_ptr(), _p(dres),
                 _dt(y), rows, c, dz_ld, _stream())
        return dy, dgamma, dbeta, dres, None, None, None, None, None, None, None, None, None, None, None, None


fuse_bn_pool = os.environ.get("URESNET_BN_POOL", "1") != "0"    # False: the max-pool after an encoder stage reads the stage's result back
fuse_bn_out_stats = os.environ.get("URESNET_BN_OUT_STATS", "1") != "0"   # False: a BatchNorm over a BatchNorm+ReLU result reads it back for its sums


def bn_act(y, bn, residual=None, stats=None, act=_lib.ACT_RELU, out=None, pool=False, out_stats=False):
    """BatchNorm2d module `bn` applied to y, fused with an optional residual add and ReLU.  `out`: optional
    channel window (alias_window) to write the result into.  pool: the caller hands the result to maxpool2x2_skip next;
    where the fused kernel covers the shape the pooled tensor is computed in the same pass and travels with the result
    (attribute `_ur_pooled`).  out_stats: the caller normalises the result with another BatchNorm next
    (bn_relu_conv1x1_pair); where the fused kernel covers the shape the per-channel sums of the result travel with it
    (attribute `_ur_out_stats`)."""
    training = bn.training or bn.running_mean is None
    if training and bn.momentum is None:
        raise NotImplementedError("BatchNorm2d(momentum=None) (cumulative average) is not implemented")
    track = training and bn.track_running_stats and bn.running_mean is not None
    rec = _BNRecord() if (fuse_bn_bwd_sums and training and torch.is_grad_enabled()) else None
    hold = [] if (pool and fuse_bn_pool and residual is None and y.is_cuda and y.dim() == 4 and torch.is_grad_enabled()) else None
    ohold = [] if (out_stats and fuse_bn_out_stats and hold is None and residual is None and out is None and y.is_cuda
                   and y.dim() == 4) else None
    z = _BNAct.apply(y, bn.weight, bn.bias, residual, stats if training else None,
                     bn.running_mean if (track or not training) else None,
                     bn.running_var if (track or not training) else None,
                     bn.num_batches_tracked if track else None,
                     training, 0.0 if bn.momentum is None else bn.momentum, bn.eps, act,
                     None if out is None else (out,), rec, hold, ohold)
    if rec is not None and getattr(rec, "y", None) is not None:
        z._ur_bn = rec           # a following 3x3 convolution picks it up (see _BNRecord)
    if hold:
        z._ur_pooled = hold[0] + (z.data_ptr(), z._version)
    if ohold:
        z._ur_out_stats = (ohold[0], z.data_ptr(), z._version)
    return z


fuse_bn_into_conv1x1 = os.environ.get("URESNET_BN_PW", "1") != "0"   # False: BatchNorm+ReLU and the 1x1 convolution stay separate


def _pw_descs(shape, cout, dtype, in_ld, out_ld):
    """(forward descriptor, backward descriptor with the produced channels widened to a multiple of 8) of a 1x1 convolution."""
    d = _desc(shape, cout, 1, 1, 1, 0, dtype)
    d.in_ld, d.out_ld = in_ld, out_ld
    db = _desc(shape, _pad8(cout), 1, 1, 1, 0, dtype)
    db.in_ld = in_ld
    return d, db


class _BNReluConv1x1(Function):
    """conv1x1(relu(BatchNorm(y))) (+bias) without the normalised tensor: the streaming 1x1 kernels apply the folded
    BatchNorm (one multiply-add per element, then ReLU) to every piece of y on its way into their LDS rows -- forward
    (ur_conv1x1_pw_fwd_affine) and backward-weight (ur_conv1x1_pw_wgrad_affine), the two kernels that read the activated
    tensor.  Backward-data yields the gradient of the activated tensor; the BatchNorm's backward sums come out of the
    backward-weight kernel (ur_conv1x1_pw_wgrad_bn: it multiplies the gradient with the ReLU mask and with mask * xhat, from
    which both the weight gradient and the sums follow) and one apply pass (mask rebuilt from y) produces dy.  Saves one pass
    over the tensor in forward (read y, write a), the reduction pass in backward, and the tensor itself."""

    @staticmethod
    def forward(ctx, y, gamma, beta, stats, running_mean, running_var, nbt, training, momentum, eps, weight, bias, out):
        _require_gpu(y)
        n, c, h, w = y.shape
        rows, dev = n * h * w, y.device
        cout = weight.shape[0]
        mean = torch.empty(c, dtype=torch.float32, device=dev)
        invstd = torch.empty(c, dtype=torch.float32, device=dev)
        aff = torch.empty(2 * c, dtype=torch.float32, device=dev)
        if training:
            if stats is None:
                stats = torch.empty((query("ur_bn_stats_rows", rows, c), c, 2), dtype=torch.float32, device=dev)
                call("ur_bn_stats", y.data_ptr(), _dt(y), rows, c, stats.data_ptr(), _stream())
            call("ur_bn_finalize_affine", stats.data_ptr(), stats.shape[0], rows, c, eps, momentum, mean.data_ptr(),
                 invstd.data_ptr(), _p(running_mean), _p(running_var), _p(nbt), gamma.data_ptr(), beta.data_ptr(),
                 aff.data_ptr(), _stream())
        else:
            call("ur_bn_eval_prepare", running_mean.data_ptr(), running_var.data_ptr(), c, eps, mean.data_ptr(),
                 invstd.data_ptr(), _stream())
            call("ur_bn_eval_affine", running_mean.data_ptr(), running_var.data_ptr(), gamma.data_ptr(), beta.data_ptr(), c, eps,
                 aff.data_ptr(), aff.data_ptr() + 4 * c, _stream())
        if out is None:
            z, out_ld = empty_nhwc(n, cout, h, w, y.dtype, dev), cout
        else:
            z, out_ld = out[0], _channel_pitch(out[0])
            if tuple(z.shape) != (n, cout, h, w) or z.dtype != y.dtype or not out_ld:
                raise RuntimeError("bn_relu_conv1x1: `out` must be a channel window with the output's shape and dtype")
        d, dbk = _pw_descs(y.shape, cout, y.dtype, c, out_ld)
        with _Timed("pw_fwd", _conv_flops(d), d):
            call("ur_conv1x1_pw_fwd_affine", y.data_ptr(), aff.data_ptr(), 1, weight.data_ptr(), cout, c, _p(bias), z.data_ptr(),
                 C.byref(d), _stream())
        ctx.save_for_backward(y, weight, gamma, beta, mean, invstd, aff)
        ctx.dbk, ctx.bias, ctx.eval = dbk, bias, not training
        return z

    @staticmethod
    def backward(ctx, dz):
        y, weight, gamma, beta, mean, invstd, aff = ctx.saved_tensors
        d = ctx.dbk
        n, c, h, w = y.shape
        rows, dev = n * h * w, y.device
        cout = weight.shape[0]
        # a channel window of a wider gradient (the backward of the Feature Merge) is read in place
        dz_ld = 0
        if cout % 8 == 0 and dz.is_cuda and dz.dtype == y.dtype and not _is_nhwc_dense(dz) and dz.data_ptr() % 16 == 0:
            dz_ld = _channel_pitch(dz)
            if dz_ld % 8:
                dz_ld = 0
        db = dw = None
        pre = _padded_head_grad(dz, d.cout) if d.cout != cout else None
        if pre is not None:                  # the loss wrote the gradient widened to d.cout channels and summed it per class
            dz, dz_ld = pre[0], d.cout
            if ctx.bias is not None and ctx.needs_input_grad[11]:
                db = grad_buffer(ctx.bias)
                db.copy_(pre[1])
        else:
            if not dz_ld:
                dz, dz_ld = _g(dz), cout
            if ctx.bias is not None and ctx.needs_input_grad[11]:
                db = _colsum(dz.data_ptr(), d.dtype, rows, cout, dz_ld, dev, ctx.bias)
            if d.cout != cout:
                dz, dz_ld = _pad_channels(dz, d.cout), d.cout
        d.out_ld = dz_ld
        fused_apply = fuse_head_bn_sums and fuse_head_bn_apply and not ctx.eval
        if not fused_apply:
            # gradient of the activated tensor a = relu(BatchNorm(y)), which exists in no buffer
            da = empty_nhwc(n, c, h, w, y.dtype, dev)
            with _Timed("pw_dgrad", _conv_flops(d), d):
                call("ur_conv1x1_pw_dgrad", dz.data_ptr(), weight.data_ptr(), cout, c, da.data_ptr(), C.byref(d), _stream())
        tb = y.numel() * y.element_size()
        sums = torch.empty((c, 2), dtype=torch.float32, device=dev)
        dgamma, dbeta = grad_buffer(gamma), grad_buffer(beta)
        if fuse_head_bn_sums and not ctx.eval:
            # weight gradient and the BatchNorm's backward sums from one kernel (see _BNReluConv1x1Pair.backward)
            if ctx.needs_input_grad[10]:
                dw = grad_buffer(weight)
            ws = _workspace(query("ur_conv1x1_pw_wgrad_bn_workspace", C.byref(d)), dev)
            with _Timed("pw_wgrad", 2 * _conv_flops(d), d):
                call("ur_conv1x1_pw_wgrad_bn", y.data_ptr(), aff.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(),
                     beta.data_ptr(), dz.data_ptr(), weight.data_ptr(), cout, _p(dw), sums.data_ptr(), dgamma.data_ptr(),
                     dbeta.data_ptr(), ws.data_ptr(), ws.numel(), C.byref(d), _stream())
        else:
            if ctx.needs_input_grad[10]:
                dw = grad_buffer(weight, (d.cout, c, 1, 1))
                ws = _workspace(query("ur_conv1x1_pw_wgrad_workspace", C.byref(d)), dev)
                with _Timed("pw_wgrad", _conv_flops(d), d):
                    call("ur_conv1x1_pw_wgrad_affine", y.data_ptr(), aff.data_ptr(), 1, dz.data_ptr(), dw.data_ptr(), ws.data_ptr(),
                         ws.numel(), C.byref(d), _stream())
                if d.cout != cout:
                    dw = dw[:cout].contiguous()
            partial = torch.empty((query("ur_bn_stats_rows", rows, c), c, 2), dtype=torch.float32, device=dev)
            with _Timed("hbm:bn_bwd_reduce", tb * 2):
                call("ur_bn_act_bwd_reduce", da.data_ptr(), y.data_ptr(), None, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(),
                     beta.data_ptr(), _lib.ACT_RELU, _dt(y), rows, c, c, partial.data_ptr(), _stream())
            call("ur_bn_bwd_finalize", partial.data_ptr(), partial.shape[0], c, sums.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                 _stream())
        dy = torch.empty_like(y, memory_format=CL)
        if fused_apply:                      # the BatchNorm's second pass in the backward-data kernel's epilogue (see the pair)
            with _Timed("pw_dgrad", _conv_flops(d), d):
                call("ur_conv1x1_pw_dgrad_bn_pair", dz.data_ptr(), weight.data_ptr(), cout, c, y.data_ptr(), dy.data_ptr(), c, c,
                     None, None, 0, 0, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), sums.data_ptr(),
                     C.byref(d), _stream())
            return dy, dgamma, dbeta, None, None, None, None, None, None, None, dw, db, None
        with _Timed("hbm:bn_bwd_apply", tb * 3):
            call("ur_bn_act_bwd_apply", da.data_ptr(), y.data_ptr(), None, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(),
                 beta.data_ptr(), sums.data_ptr(), _lib.ACT_RELU, int(ctx.eval), dy.data_ptr(), None, _dt(y), rows, c, c,
                 _stream())
        return dy, dgamma, dbeta, None, None, None, None, None, None, None, dw, db, None


def bn_relu_conv1x1(y, bn, conv, stats=None, out=None):
    """conv(relu(bn(y))) for a BatchNorm2d `bn` and a 1x1 stride-1 Conv2d `conv`, fused as _BNReluConv1x1 where the streaming
    kernels cover the shape in all three directions; None otherwise (the caller then runs the two modules separately)."""
    weight, bias = conv.weight, conv.bias
    if (not (fuse_bn_into_conv1x1 and use_pw_kernel) or not y.is_cuda or y.dim() != 4 or y.dtype != torch.bfloat16
            or not _is_nhwc_dense(y) or y.data_ptr() % 16 or tuple(weight.shape[2:]) != (1, 1) or weight.dtype != torch.float32
            or not weight.is_contiguous() or weight.shape[1] != y.shape[1] or y.shape[1] % 8):
        return None
    training = bn.training or bn.running_mean is None
    if training and bn.momentum is None:
        return None
    cout = weight.shape[0]
    out_ld = cout if out is None else _channel_pitch(out)
    if not out_ld:
        return None
    d, dbk = _pw_descs(y.shape, cout, y.dtype, y.shape[1], out_ld)
    dbk.out_ld = dbk.cout
    if not (query("ur_conv1x1_pw_supported", C.byref(d), 0) and query("ur_conv1x1_pw_supported", C.byref(dbk), 1)
            and query("ur_conv1x1_pw_wgrad_supported", C.byref(dbk))):
        return None
    track = training and bn.track_running_stats and bn.running_mean is not None
    return _BNReluConv1x1.apply(y, bn.weight, bn.bias, stats if training else None,
                                bn.running_mean if (track or not training) else None,
                                bn.running_var if (track or not training) else None,
                                bn.num_batches_tracked if track else None,
                                training, 0.0 if bn.momentum is None else bn.momentum, bn.eps, weight, bias,
                                None if out is None else (out,))


class _BNReluConv1x1Pair(Function):
    """_BNReluConv1x1 over a channel concatenation that is never materialised: conv1x1(relu(BatchNorm(cat([ya, yb], 1)))).
    The BatchNorm is per channel, so its statistics, constants and backward passes run on the two dense tensors separately
    (slices of the parameter vectors); the streaming 1x1 kernels read both tensors (ur_conv1x1_pw_*_pair).  A 64-channel
    window of an 80-channel tensor is 128-byte runs at a 160-byte pitch -- partial memory lines in every pass that touches it;
    two dense tensors stream at full rate."""

    @staticmethod
    def forward(ctx, ya, yb, gamma, beta, running_mean, running_var, nbt, training, momentum, eps, weight, bias, stats_a=None):
        _require_gpu(ya)
        n, ca, h, w = ya.shape
        cb = yb.shape[1]
        c = ca + cb
        rows, dev = n * h * w, ya.device
        cout = weight.shape[0]
        mean = torch.empty(c, dtype=torch.float32, device=dev)
        invstd = torch.empty(c, dtype=torch.float32, device=dev)
        affs = (torch.empty(2 * ca, dtype=torch.float32, device=dev), torch.empty(2 * cb, dtype=torch.float32, device=dev))
        parts = ((ya, 0, ca, affs[0]), (yb, ca, cb, affs[1]))
        off = lambda t, o: None if t is None else t.data_ptr() + 4 * o      # noqa: E731  (fp32 vectors: element o)
        if training:
            for y, o, ci, aff in parts:
                stats = stats_a if o == 0 else None          # (the producer of ya summed its result: bn_act(out_stats=True))
                if stats is None:
                    stats = torch.empty((query("ur_bn_stats_rows", rows, ci), ci, 2), dtype=torch.float32, device=dev)
                    call("ur_bn_stats", y.data_ptr(), _dt(y), rows, ci, stats.data_ptr(), _stream())
                call("ur_bn_finalize_affine", stats.data_ptr(), stats.shape[0], rows, ci, eps, momentum, off(mean, o), off(invstd, o),
                     off(running_mean, o), off(running_var, o), _p(nbt) if o == 0 else None, off(gamma, o), off(beta, o),
                     aff.data_ptr(), _stream())
        else:
            call("ur_bn_eval_prepare", running_mean.data_ptr(), running_var.data_ptr(), c, eps, mean.data_ptr(),
                 invstd.data_ptr(), _stream())
            for y, o, ci, aff in parts:
                call("ur_bn_eval_affine", off(running_mean, o), off(running_var, o), off(gamma, o), off(beta, o), ci, eps,
                     aff.data_ptr(), aff.data_ptr() + 4 * ci, _stream())
        z = empty_nhwc(n, cout, h, w, ya.dtype, dev)
        d, dbk = _pw_descs((n, c, h, w), cout, ya.dtype, c, cout)
        with _Timed("pw_fwd", _conv_flops(d), d):
            call("ur_conv1x1_pw_fwd_affine_pair", ya.data_ptr(), ca, ca, yb.data_ptr(), cb, cb, affs[0].data_ptr(),
                 affs[1].data_ptr(), 1, weight.data_ptr(), cout, c, _p(bias), z.data_ptr(), C.byref(d), _stream())
        ctx.save_for_backward(ya, yb, weight, gamma, beta, mean, invstd, affs[0], affs[1])
        ctx.dbk, ctx.bias, ctx.eval = dbk, bias, not training
        return z

    @staticmethod
    def backward(ctx, dz):
        ya, yb, weight, gamma, beta, mean, invstd, aff_a, aff_b = ctx.saved_tensors
        d = ctx.dbk
        n, ca, h, w = ya.shape
        cb = yb.shape[1]
        c = ca + cb
        rows, dev = n * h * w, ya.device
        cout = weight.shape[0]
        db = dw = None
        pre = _padded_head_grad(dz, d.cout) if d.cout != cout else None
        if pre is not None:                  # the loss wrote the gradient widened to d.cout channels and summed it per class
            dz, dz_ld = pre[0], d.cout
            if ctx.bias is not None and ctx.needs_input_grad[11]:
                db = grad_buffer(ctx.bias)
                db.copy_(pre[1])
        else:
            dz, dz_ld = _g(dz), cout
            if ctx.bias is not None and ctx.needs_input_grad[11]:
                db = _colsum(dz.data_ptr(), d.dtype, rows, cout, dz_ld, dev, ctx.bias)
            if d.cout != cout:
                dz, dz_ld = _pad_channels(dz, d.cout), d.cout
        d.out_ld = dz_ld
        dgamma, dbeta = grad_buffer(gamma), grad_buffer(beta)
        off = lambda t, o: t.data_ptr() + 4 * o      # noqa: E731
        all_sums = None
        fused_apply = fuse_head_bn_sums and fuse_head_bn_apply and not ctx.eval
        if not fused_apply:
            # gradients of the two halves of the activated tensor a = relu(BatchNorm(cat)), which exists in no buffer
            da = (empty_nhwc(n, ca, h, w, ya.dtype, dev), empty_nhwc(n, cb, h, w, ya.dtype, dev))
            with _Timed("pw_dgrad", _conv_flops(d), d):
                call("ur_conv1x1_pw_dgrad_pair", dz.data_ptr(), weight.data_ptr(), cout, c, da[0].data_ptr(), ca, ca, da[1].data_ptr(),
                     cb, cb, C.byref(d), _stream())
        if fuse_head_bn_sums and not ctx.eval:
            # the backward-weight kernel multiplies the gradient with the ReLU mask and with mask * xhat: the weight gradient
            # and the BatchNorm's backward sums both follow from those two products (ur_conv1x1_pw_wgrad_bn_pair), and the
            # reduction passes over (da, ya) and (db, yb) are not needed
            all_sums = torch.empty((c, 2), dtype=torch.float32, device=dev)
            if ctx.needs_input_grad[10]:
                dw = grad_buffer(weight)
            ws = _workspace(query("ur_conv1x1_pw_wgrad_bn_workspace", C.byref(d)), dev)
            with _Timed("pw_wgrad", 2 * _conv_flops(d), d):
                call("ur_conv1x1_pw_wgrad_bn_pair", ya.data_ptr(), ca, ca, yb.data_ptr(), cb, cb, aff_a.data_ptr(), aff_b.data_ptr(),
                     mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), dz.data_ptr(), weight.data_ptr(), cout,
                     _p(dw), all_sums.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(), ws.numel(), C.byref(d),
                     _stream())
        elif ctx.needs_input_grad[10]:
            dw = grad_buffer(weight, (d.cout, c, 1, 1))
            ws = _workspace(query("ur_conv1x1_pw_wgrad_workspace", C.byref(d)), dev)
            with _Timed("pw_wgrad", _conv_flops(d), d):
                call("ur_conv1x1_pw_wgrad_affine_pair", ya.data_ptr(), ca, ca, yb.data_ptr(), cb, cb, aff_a.data_ptr(),
                     aff_b.data_ptr(), 1, dz.data_ptr(), dw.data_ptr(), ws.data_ptr(), ws.numel(), C.byref(d), _stream())
            if d.cout != cout:
                dw = dw[:cout].contiguous()
        if fused_apply:
            # the sums are known before the backward-data kernel runs, so the BatchNorm's second pass moves into its epilogue:
            # the gradient of the activated tensor is neither written nor read back
            dys = [torch.empty_like(ya, memory_format=CL), torch.empty_like(yb, memory_format=CL)]
            with _Timed("pw_dgrad", _conv_flops(d), d):
                call("ur_conv1x1_pw_dgrad_bn_pair", dz.data_ptr(), weight.data_ptr(), cout, c, ya.data_ptr(), dys[0].data_ptr(), ca, ca,
                     yb.data_ptr(), dys[1].data_ptr(), cb, cb, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                     all_sums.data_ptr(), C.byref(d), _stream())
            return dys[0], dys[1], dgamma, dbeta, None, None, None, None, None, None, dw, db, None
        dys = []
        for y, g, o, ci in ((ya, da[0], 0, ca), (yb, da[1], ca, cb)):
            tb = y.numel() * y.element_size()
            if all_sums is not None:
                sums = all_sums[o:o + ci]
            else:
                partial = torch.empty((query("ur_bn_stats_rows", rows, ci), ci, 2), dtype=torch.float32, device=dev)
                with _Timed("hbm:bn_bwd_reduce", tb * 2):
                    call("ur_bn_act_bwd_reduce", g.data_ptr(), y.data_ptr(), None, off(mean, o), off(invstd, o), off(gamma, o),
                         off(beta, o), _lib.ACT_RELU, _dt(y), rows, ci, ci, partial.data_ptr(), _stream())
                sums = torch.empty((ci, 2), dtype=torch.float32, device=dev)
                call("ur_bn_bwd_finalize", partial.data_ptr(), partial.shape[0], ci, sums.data_ptr(), off(dgamma, o), off(dbeta, o),
                     _stream())
            dy = torch.empty_like(y, memory_format=CL)
            with _Timed("hbm:bn_bwd_apply", tb * 3):
                call("ur_bn_act_bwd_apply", g.data_ptr(), y.data_ptr(), None, off(mean, o), off(invstd, o), off(gamma, o),
                     off(beta, o), sums.data_ptr(), _lib.ACT_RELU, int(ctx.eval), dy.data_ptr(), None, _dt(y), rows, ci, ci,
                     _stream())
            dys.append(dy)
        return dys[0], dys[1], dgamma, dbeta, None, None, None, None, None, None, dw, db, None


fuse_merge_into_head = os.environ.get("URESNET_MERGE_PAIR", "1") != "0"    # False: the Feature Merge builds the merged tensor
fuse_head_bn_sums = os.environ.get("URESNET_HEAD_BN_SUMS", "1") != "0"    # False: the head's BatchNorm backward reduces over its input gradient
fuse_head_bn_apply = os.environ.get("URESNET_HEAD_BN_APPLY", "1") != "0"   # False: ... and applies in a pass of its own (needs the sums fused)


def bn_relu_conv1x1_pair_supported(n, ca, cb, h, w, dtype, bn, conv):
    """Whether bn_relu_conv1x1_pair covers conv(relu(bn(cat))) for two dense bf16 tensors of ca and cb channels."""
    weight = conv.weight
    c = ca + cb
    if not (fuse_merge_into_head and fuse_bn_into_conv1x1 and use_pw_kernel) or dtype != torch.bfloat16 or ca % 8 or cb % 8:
        return False
    if (tuple(weight.shape[2:]) != (1, 1) or weight.dtype != torch.float32 or not weight.is_contiguous() or weight.shape[1] != c
            or bn.num_features != c or not weight.is_cuda):
        return False
    if (bn.training or bn.running_mean is None) and bn.momentum is None:
        return False
    cout = weight.shape[0]
    d, dbk = _pw_descs((n, c, h, w), cout, dtype, c, cout)
    dbk.out_ld = dbk.cout
    return bool(query("ur_conv1x1_pw_supported", C.byref(d), 0) and query("ur_conv1x1_pw_supported", C.byref(dbk), 1)
                and query("ur_conv1x1_pw_wgrad_supported", C.byref(dbk)))


def bn_relu_conv1x1_pair(ya, yb, bn, conv):
    """conv(relu(bn(cat([ya, yb], 1)))) for a BatchNorm2d `bn` over both tensors' channels and a 1x1 stride-1 Conv2d `conv`,
    without the concatenated tensor (_BNReluConv1x1Pair); None where the streaming kernels do not cover the shape (the caller
    then concatenates)."""
    for y in (ya, yb):
        if not y.is_cuda or y.dim() != 4 or not _is_nhwc_dense(y) or y.data_ptr() % 16:
            return None
    if (tuple(ya.shape[2:]) != tuple(yb.shape[2:]) or ya.shape[0] != yb.shape[0] or ya.dtype != yb.dtype
            or not bn_relu_conv1x1_pair_supported(ya.shape[0], ya.shape[1], yb.shape[1], ya.shape[2], ya.shape[3], ya.dtype, bn, conv)):
        return None
    training = bn.training or bn.running_mean is None
    track = training and bn.track_running_stats and bn.running_mean is not None
    pre = getattr(ya, "_ur_out_stats", None)
    stats_a = None
    if pre is not None:
        ya._ur_out_stats = None
        if training and pre[1] == ya.data_ptr() and pre[2] == ya._version and tuple(pre[0].shape[1:]) == (ya.shape[1], 2):
            stats_a = pre[0]                             # (sum, sum of squares) of ya from the pass that wrote it, unmodified since
    return _BNReluConv1x1Pair.apply(ya, yb, bn.weight, bn.bias,
                                    bn.running_mean if (track or not training) else None,
                                    bn.running_var if (track or not training) else None,
                                    bn.num_batches_tracked if track else None,
                                    training, 0.0 if bn.momentum is None else bn.momentum, bn.eps, conv.weight, conv.bias, stats_a)


# ------------------------------------------------------------------------------ spatial ops
class _MaxPool2x2(Function):
    @staticmethod
    def forward(ctx, x):
        _require_gpu(x)
        n, c, h, w = x.shape
        y = empty_nhwc(n, c, h // 2, w // 2, x.dtype, x.device)
        idx = torch.empty((n, h // 2, w // 2, c), dtype=torch.uint8, device=x.device)
        with _Timed("hbm:maxpool_fwd", x.numel() * x.element_size() + y.numel() * (y.element_size() + 1)):
            call("ur_maxpool2x2_fwd", x.data_ptr(), y.data_ptr(), idx.data_ptr(), _dt(x), n, h, w, c, _channel_pitch(x) or c,
                 _stream())
        ctx.save_for_backward(idx)
        ctx.shape = (n, c, h, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        n, c, h, w = ctx.shape
        dy = _g(dy)
        dx = empty_nhwc(n, c, h, w, dy.dtype, dy.device)
        with _Timed("hbm:maxpool_bwd", dx.numel() * dx.element_size() + dy.numel() * (dy.element_size() + 1)):
            call("ur_maxpool2x2_bwd", dy.data_ptr(), idx.data_ptr(), dx.data_ptr(), None, 0, _dt(dy), n, h, w, c, _stream())
        return dx


def maxpool2x2(x):
    return _MaxPool2x2.apply(x)


class _MaxPool2x2Skip(Function):
    """x -> (maxpool2x2(x), x): the second output aliases x and goes to the skip connection, so that both gradients of
    x arrive in one backward call and are summed inside the pooling backward kernel (no separate add pass)."""

    @staticmethod
    def forward(ctx, x):
        _require_gpu(x)
        n, c, h, w = x.shape
        pre = getattr(x, "_ur_pooled", None)
        if pre is not None:
            x._ur_pooled = None
        if pre is not None and pre[2] == x.data_ptr() and pre[3] == x._version and tuple(pre[0].shape) == (n, c, h // 2, w // 2):
            y, idx = pre[0], pre[1]                  # the BatchNorm pass that wrote x pooled it too (bn_act(pool=True))
        else:
            y = empty_nhwc(n, c, h // 2, w // 2, x.dtype, x.device)
            idx = torch.empty((n, h // 2, w // 2, c), dtype=torch.uint8, device=x.device)
            with _Timed("hbm:maxpool_fwd", x.numel() * x.element_size() + y.numel() * (y.element_size() + 1)):
                call("ur_maxpool2x2_fwd", x.data_ptr(), y.data_ptr(), idx.data_ptr(), _dt(x), n, h, w, c, _channel_pitch(x) or c,
                     _stream())
        ctx.save_for_backward(idx)
        ctx.shape = (n, c, h, w)
        ctx.bn_rec = getattr(x, "_ur_bn", None)      # x is a BatchNorm+ReLU output: its backward sums can come from here
        ctx.set_materialize_grads(False)
        skip = torch.empty(0, dtype=x.dtype, device=x.device)
        skip.set_(x.untyped_storage(), x.storage_offset(), x.size(), x.stride())
        return y, skip

    @staticmethod
    def backward(ctx, dy, dskip):
        (idx,) = ctx.saved_tensors
        n, c, h, w = ctx.shape
        if dy is None:
            return dskip
        dy = _g(dy)
        add_ld = 0
        if dskip is not None:
            add_ld = _channel_pitch(dskip) if (dskip.is_cuda and dskip.dtype == dy.dtype) else 0
            if not add_ld:
                dskip, add_ld = _g(dskip), c
        dx = empty_nhwc(n, c, h, w, dy.dtype, dy.device)
        tb = dx.numel() * dx.element_size()
        rec = ctx.bn_rec
        if (fuse_bn_bwd_sums and fuse_pool_bn_sums and rec is not None and not rec.use_z and rec.y.dtype == dy.dtype
                and tuple(rec.y.shape) == (n, c, h, w) and _is_nhwc_dense(rec.y) and (dskip is None or add_ld % 8 == 0)
                and (dskip is None or dskip.data_ptr() % 16 == 0)
                and query("ur_maxpool2x2_bwd_bnsums_supported", _dt(dy), n, h, w, c)):
            # dx is the gradient of a BatchNorm+ReLU output: the kernel that forms it also emits that BatchNorm's backward sums
            partial = torch.empty((query("ur_bn_stats_rows", n * h * w, c), c, 2), dtype=torch.float32, device=dy.device)
            with _Timed("hbm:maxpool_bwd_bn", tb * (3 if dskip is not None else 2) + dy.numel() * (dy.element_size() + 1)):
                call("ur_maxpool2x2_bwd_bnsums", dy.data_ptr(), idx.data_ptr(), dx.data_ptr(), _p(dskip), add_ld,
                     rec.y.data_ptr(), rec.mean.data_ptr(), rec.invstd.data_ptr(), rec.gamma.data_ptr(), rec.beta.data_ptr(),
                     partial.data_ptr(), _dt(dy), n, h, w, c, _stream())
            rec.partial, rec.dz_ptr, rec.dz_version = partial, dx.data_ptr(), dx._version
            return dx
        with _Timed("hbm:maxpool_bwd", tb * (2 if dskip is not None else 1) + dy.numel() * (dy.element_size() + 1)):
            call("ur_maxpool2x2_bwd", dy.data_ptr(), idx.data_ptr(), dx.data_ptr(), _p(dskip), add_ld, _dt(dy), n, h, w, c,
                 _stream())
        return dx


def maxpool2x2_skip(x):
    """(maxpool2x2(x), alias of x for the skip connection); x: dense NHWC or a channel window."""
    return _MaxPool2x2Skip.apply(x)


class _UpsampleBilinear(Function):
    @staticmethod
    def forward(ctx, x, scale):
        _require_gpu(x)
        n, c, h, w = x.shape
        y = empty_nhwc(n, c, h * scale, w * scale, x.dtype, x.device)
        with _Timed("hbm:upsample_fwd", (x.numel() + y.numel()) * x.element_size()):
            call("ur_upsample_bilinear_fwd", x.data_ptr(), y.data_ptr(), _dt(x), n, h, w, c, scale, _stream())
        ctx.shape, ctx.scale = (n, c, h, w), scale
        return y

    @staticmethod
    def backward(ctx, dy):
        n, c, h, w = ctx.shape
        dy = _g(dy)
        dx = empty_nhwc(n, c, h, w, dy.dtype, dy.device)
        with _Timed("hbm:upsample_bwd", (dx.numel() + dy.numel()) * dy.element_size()):
            call("ur_upsample_bilinear_bwd", dy.data_ptr(), dx.data_ptr(), _dt(dy), n, h, w, c, ctx.scale, _stream())
        return dx, None


def upsample_bilinear(x, scale):
    if int(scale) != scale or scale < 1:
        raise NotImplementedError("bilinear upsampling is implemented for integer scale factors only (got %r)" % scale)
    return _UpsampleBilinear.apply(x, int(scale))


class _AvgPool(Function):
    @staticmethod
    def forward(ctx, x, k):
        _require_gpu(x)
        n, c, h, w = x.shape
        y = empty_nhwc(n, c, h // k, w // k, x.dtype, x.device)
        call("ur_avgpool_fwd", x.data_ptr(), y.data_ptr(), _dt(x), n, h, w, c, k, _stream())
        ctx.shape, ctx.k = (n, c, h, w), k
        return y

    @staticmethod
    def backward(ctx, dy):
        n, c, h, w = ctx.shape
        dy = _g(dy)
        dx = empty_nhwc(n, c, h, w, dy.dtype, dy.device)
        call("ur_avgpool_bwd", dy.data_ptr(), dx.data_ptr(), _dt(dy), n, h, w, c, ctx.k, _stream())
        return dx, None


def avg_pool2d(x, k):
    return _AvgPool.apply(x, int(k))


def _copy_channels(src, src_coff, c, dst, dst_coff, pad_t=0, pad_l=0):
    n, cs, sh, sw = src.shape
    _, cd, dh, dw = dst.shape
    call("ur_copy_channels", src.data_ptr(), cs, src_coff, sh, sw, dst.data_ptr(), cd, dst_coff, dh, dw, pad_t, pad_l,
         n, c, _dt(src), _stream())


class _CatPad(Function):
    """torch.cat([a, pad(b)], dim=1): b is zero-padded (or cropped) to a's spatial size first (S/C:119-124)."""

    @staticmethod
    def forward(ctx, a, b):
        _require_gpu(a)
        n, ca, h, w = a.shape
        _, cb, hb, wb = b.shape
        out = empty_nhwc(n, ca + cb, h, w, a.dtype, a.device)
        pt, pl = (h - hb) // 2, (w - wb) // 2
        _copy_channels(a, 0, ca, out, 0)
        _copy_channels(b, 0, cb, out, ca, pt, pl)
        ctx.meta = (ca, cb, hb, wb, pt, pl)
        return out

    @staticmethod
    def backward(ctx, g):
        ca, cb, hb, wb, pt, pl = ctx.meta
        g = _g(g)
        n, _, h, w = g.shape
        ga = gb = None
        if ctx.needs_input_grad[0]:
            ga = empty_nhwc(n, ca, h, w, g.dtype, g.device)
            _copy_channels(g, 0, ca, ga, 0)
        if ctx.needs_input_grad[1]:
            gb = empty_nhwc(n, cb, hb, wb, g.dtype, g.device)
            _copy_channels(g, ca, cb, gb, 0, -pt, -pl)
        return ga, gb


def cat_pad(a, b):
    return _CatPad.apply(a, b)


def cat_channels(a, b):
    """torch.cat([a, b], dim=1) with torch's own size check and message (the classification heads of the reference
    concatenate without padding: mismatched feature maps are an error there, not something to paper over)."""
    if a.dim() != 4 or b.dim() != 4:
        raise RuntimeError("cat_channels: 4-D tensors expected")
    for dim in (0, 2, 3):
        if a.shape[dim] != b.shape[dim]:
            raise RuntimeError("Sizes of tensors must match except in dimension 1. Expected size %d but got size %d for "
                               "tensor number 1 in the list." % (a.shape[dim], b.shape[dim]))
    return _CatPad.apply(a, b)


class _CatWindows(Function):
    """torch.cat([a, b], dim=1) where a and b already ARE the two channel windows of `buf` (their producers wrote
    them in place, see alias_window): nothing is copied in either direction, the gradients are channel windows of
    the incoming gradient."""

    @staticmethod
    def forward(ctx, a, b, buf):
        ctx.ca = a.shape[1]
        return buf[0]

    @staticmethod
    def backward(ctx, g):
        g = _g(g)
        return (g[:, :ctx.ca] if ctx.needs_input_grad[0] else None,
                g[:, ctx.ca:] if ctx.needs_input_grad[1] else None, None)


def cat_windows(a, b, buf):
    """cat([a, b], dim=1) for a = alias_window(buf, 0, ca), b = alias_window(buf, ca, cb), both already written."""
    ca, cb = a.shape[1], b.shape[1]
    es = buf.element_size()
    if (buf.shape[1] != ca + cb or a.data_ptr() != buf.data_ptr() or b.data_ptr() != buf.data_ptr() + ca * es
            or _channel_pitch(a) != ca + cb or _channel_pitch(b) != ca + cb):
        raise RuntimeError("cat_windows: a and b must be the two channel windows of buf")
    return _CatWindows.apply(a, b, (buf,))


class _ChannelShuffle(Function):
    @staticmethod
    def forward(ctx, x, groups):
        n, c, h, w = x.shape
        y = torch.empty_like(x, memory_format=CL)
        call("ur_channel_shuffle", x.data_ptr(), y.data_ptr(), _dt(x), n * h * w, c, groups, _stream())
        ctx.groups = groups
        return y

    @staticmethod
    def backward(ctx, g):
        g = _g(g)
        n, c, h, w = g.shape
        dx = torch.empty_like(g, memory_format=CL)
        # the inverse permutation of a shuffle with `groups` groups is a shuffle with c/groups groups
        call("ur_channel_shuffle", g.data_ptr(), dx.data_ptr(), _dt(g), n * h * w, c, c // ctx.groups, _stream())
        return dx, None


def channel_shuffle(x, groups):
    """Reference S/C:6-15.  groups == C (every call site in U-ResNet) or groups == 1 is the identity
    permutation: the input is returned unchanged (bit-exact, no copy)."""
    c = x.shape[1]
    if c % groups != 0:
        raise RuntimeError("channel_shuffle: %d channels are not divisible by %d groups" % (c, groups))
    if groups == c or groups == 1:
        return x
    return _ChannelShuffle.apply(x, groups)


class _Act(Function):
    @staticmethod
    def forward(ctx, x, act, slope):
        _require_gpu(x)
        y = torch.empty_like(x, memory_format=CL) if x.dim() == 4 else torch.empty_like(x)
        call("ur_act_fwd", x.data_ptr(), y.data_ptr(), _dt(x), x.numel(), act, slope, _stream())
        ctx.save_for_backward(x)
        ctx.act, ctx.slope = act, slope
        return y

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        g = _g(g) if g.dim() == 4 else g.contiguous()
        dx = torch.empty_like(g, memory_format=CL) if g.dim() == 4 else torch.empty_like(g)
        call("ur_act_bwd", g.data_ptr(), x.data_ptr(), dx.data_ptr(), _dt(x), x.numel(), ctx.act, ctx.slope, _stream())
        return dx, None, None


def activation(x, act, slope=0.0):
    return _Act.apply(x, act, float(slope))


class _PReLU(Function):
    """nn.PReLU: slope per channel (weight.numel() == C) or shared (1); activations are NHWC or [rows][C]."""

    @staticmethod
    def forward(ctx, x, weight):
        _require_gpu(x)
        c = x.shape[1]
        if weight.numel() not in (1, c):
            raise RuntimeError("Mismatch of parameter numbers and input channel size. Found parameter numbers = %d and channel size = %d."
                               % (weight.numel(), c))
        if x.dim() == 4:
            x = _g(x)
        y = torch.empty_like(x, memory_format=CL) if x.dim() == 4 else torch.empty_like(x)
        call("ur_prelu_fwd", x.data_ptr(), weight.data_ptr(), weight.numel(), y.data_ptr(), _dt(x), x.numel(), c, _stream())
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        c = x.shape[1]
        g = _g(g) if g.dim() == 4 else g.contiguous()
        dx = torch.empty_like(g, memory_format=CL) if g.dim() == 4 else torch.empty_like(g)
        terms = torch.empty(x.numel(), dtype=torch.float32, device=x.device)
        call("ur_prelu_bwd", g.data_ptr(), x.data_ptr(), weight.data_ptr(), weight.numel(), dx.data_ptr(), terms.data_ptr(), _dt(x),
             x.numel(), c, _stream())
        if weight.numel() == 1:
            dw = _colsum(terms.data_ptr(), _lib.UR_F32, x.numel(), 1, 1, x.device)
        else:
            dw = _colsum(terms.data_ptr(), _lib.UR_F32, x.numel() // c, c, c, x.device)
        return dx, dw.view_as(weight)


def prelu(x, weight):
    return _PReLU.apply(x, weight)


class _Add(Function):
    """out = a + b (residual add outside a fused BN epilogue)."""

    @staticmethod
    def forward(ctx, a, b):
        out = torch.empty_like(a, memory_format=torch.preserve_format)
        call("ur_nchw_to_nhwc", a.data_ptr(), _dt(a), out.data_ptr(), _dt(a), 1, 1, 1, a.numel(), _stream())
        call("ur_add_inplace", out.data_ptr(), b.data_ptr(), _dt(a), a.numel(), _stream())
        return out

    @staticmethod
    def backward(ctx, g):
        return g, g


def add(a, b):
    if a.shape != b.shape or a.stride() != b.stride():
        raise RuntimeError("add: operands must have identical shape and layout")
    return _Add.apply(a, b)


# ------------------------------------------------------------------------------ losses and masks
# CE backward -> 1x1 head: the widened gradient and its column sums, keyed by the address of the gradient tensor the loss
# returned (URESNET_CE_HEAD_GRAD=0: plain gradient, the head widens and sums it itself)
fuse_ce_head_grad = os.environ.get("URESNET_CE_HEAD_GRAD", "1") != "0"
_padded_grads = {}


def _padded_head_grad(dz, cpad):
    """(widened gradient [n, cpad, h, w] with zero fill, per-channel sums) when dz is the window the loss's backward returned
    for exactly this width, untouched since; else None."""
    hit = _padded_grads.pop(dz.data_ptr(), None) if _padded_grads else None
    if hit is None:
        return None
    buf, colsum, version = hit
    if (buf.shape[1] != cpad or dz._version != version or tuple(dz.shape[2:]) != tuple(buf.shape[2:]) or dz.shape[0] != buf.shape[0]
            or _channel_pitch(dz) != cpad or dz.dtype != buf.dtype):
        return None
    return buf, colsum


class _CrossEntropy(Function):
    @staticmethod
    def forward(ctx, logits, target, ignore_index, count_hook):
        _require_gpu(logits)
        if logits.dim() == 4:
            n, c, h, w = logits.shape
            rows = n * h * w
            if not _is_nhwc_dense(logits):
                logits = to_nhwc_raw(logits.contiguous(), logits.dtype)
        else:
            rows, c = logits.shape
            logits = logits.contiguous()
        target = target.contiguous()
        if target.dtype != torch.int64:
            raise TypeError("cross_entropy: target must be int64 (the reference calls targets.long())")
        if target.numel() != rows:
            raise RuntimeError("cross_entropy: %d targets for %d logit rows" % (target.numel(), rows))
        dev = logits.device
        lse = torch.empty(rows, dtype=torch.float32, device=dev)
        out3 = torch.empty(3, dtype=torch.float32, device=dev)
        ws = _workspace(query("ur_loss_workspace", rows), dev)
        with _Timed("hbm:ce_fwd", logits.numel() * logits.element_size() + rows * 12):      # logits, int64 target, fp32 lse
            call("ur_ce_fwd", logits.data_ptr(), _dt(logits), target.data_ptr(), rows, c, c, ignore_index, lse.data_ptr(),
                 out3.data_ptr(), ws.data_ptr(), ws.numel(), _stream())
        count = out3[1:2]
        if count_hook is not None:          # data parallel: mean over the GLOBAL number of valid pixels
            count = count_hook(out3)
        ctx.save_for_backward(logits, target, lse, count)
        ctx.meta = (rows, c, ignore_index)
        return out3[2] if count_hook is None else out3[0] / count[0]

    @staticmethod
    def backward(ctx, gout):
        logits, target, lse, count = ctx.saved_tensors
        rows, c, ignore_index = ctx.meta
        gout = gout.contiguous().float()
        cpad = _pad8(c)
        if (fuse_ce_head_grad and logits.dim() == 4 and cpad != c
                and query("ur_ce_bwd_padded_supported", _dt(logits), c, cpad)):
            # The layer behind the loss is the 1x1 head: hand it the gradient in the form its kernels read (rows widened to a
            # multiple of 8 channels, zero filled) together with the per-class sums (its bias gradient).  The result is a
            # channel window of the widened tensor; _padded_head_grad() recognises it by address.
            n, _, h, w = logits.shape
            buf = empty_nhwc(n, cpad, h, w, logits.dtype, logits.device)
            colsum = torch.empty(c, dtype=torch.float32, device=logits.device)
            ws = _workspace(query("ur_ce_bwd_padded_workspace", rows), logits.device)
            with _Timed("hbm:ce_bwd", logits.numel() * logits.element_size() + buf.numel() * buf.element_size() + rows * 12):
                call("ur_ce_bwd_padded", logits.data_ptr(), _dt(logits), target.data_ptr(), rows, c, ignore_index, lse.data_ptr(),
                     count.data_ptr(), gout.data_ptr(), buf.data_ptr(), cpad, colsum.data_ptr(), ws.data_ptr(), ws.numel(), _stream())
            dlogits = alias_window(buf, 0, c)
            _padded_grads.clear()
            _padded_grads[dlogits.data_ptr()] = (buf, colsum, dlogits._version)
            return dlogits, None, None, None
        dlogits = torch.empty_like(logits, memory_format=torch.preserve_format)
        with _Timed("hbm:ce_bwd", 2 * logits.numel() * logits.element_size() + rows * 12):
            call("ur_ce_bwd", logits.data_ptr(), _dt(logits), target.data_ptr(), rows, c, c, ignore_index, lse.data_ptr(),
                 count.data_ptr(), gout.data_ptr(), dlogits.data_ptr(), c, _stream())
        return dlogits, None, None, None


def cross_entropy(logits, target, ignore_index=-100, count_hook=None):
    return _CrossEntropy.apply(logits, target, ignore_index, count_hook)


class _BCEWithLogits(Function):
    @staticmethod
    def forward(ctx, logits, target):
        _require_gpu(logits)
        if logits.dim() == 4 and not _is_nhwc_dense(logits):
            logits = to_nhwc_raw(logits.contiguous(), logits.dtype)
        if target.shape != logits.shape:
            raise RuntimeError("bce_with_logits: target shape %s != logits shape %s" % (tuple(target.shape),
                                                                                        tuple(logits.shape)))
        # elementwise pairing must follow the logits' physical order
        if target.dim() == 4:
            target = to_nhwc_raw(target, torch.float32)
        else:
            target = target.float().contiguous()
        n = logits.numel()
        dev = logits.device
        out3 = torch.empty(3, dtype=torch.float32, device=dev)
        ws = _workspace(query("ur_loss_workspace", n), dev)
        call("ur_bce_fwd", logits.data_ptr(), _dt(logits), target.data_ptr(), n, out3.data_ptr(), ws.data_ptr(),
             ws.numel(), _stream())
        ctx.save_for_backward(logits, target)
        return out3[2]

    @staticmethod
    def backward(ctx, gout):
        logits, target = ctx.saved_tensors
        gout = gout.contiguous().float()
        dlogits = torch.empty_like(logits, memory_format=torch.preserve_format)
        call("ur_bce_bwd", logits.data_ptr(), _dt(logits), target.data_ptr(), logits.numel(), gout.data_ptr(),
             dlogits.data_ptr(), _stream())
        return dlogits, None


def bce_with_logits(logits, target):
    return _BCEWithLogits.apply(logits, target)


def confusion_matrix(logits, targets, num_classes, ignore_index=255, out=None):
    """int64 [num_classes, num_classes] counts cm[target][argmax(logits)] over the pixels whose target is a class
    (argmax and histogram in one pass; `out` accumulates).  logits: [N,C,H,W] activation or [rows, C]."""
    _require_gpu(logits)
    if logits.dim() == 4:
        n, c, h, w = logits.shape
        if not _is_nhwc_dense(logits):
            logits = to_nhwc_raw(logits.contiguous(), logits.dtype)
        rows = n * h * w
    else:
        rows, c = logits.shape
        logits = logits.contiguous()
    if c != num_classes:
        raise RuntimeError("confusion_matrix: logits have %d channels, num_classes is %d" % (c, num_classes))
    if targets.dtype != torch.int64:
        targets = targets.long()
    targets = targets.contiguous()
    if targets.numel() != rows or targets.device != logits.device:
        raise RuntimeError("confusion_matrix: targets must hold one class per pixel on the logits' device")
    cm = torch.zeros((c, c), dtype=torch.int64, device=logits.device) if out is None else out
    call("ur_confusion_matrix", logits.data_ptr(), _dt(logits), targets.data_ptr(), rows, c, c, ignore_index, cm.data_ptr(),
         _stream())
    return cm


def seg_metrics_accumulate(cm, acc, loss=None):
    """acc (float64[7], device) += (precision, recall, accuracy, f1, iou, 1, loss) of the confusion matrix cm."""
    if acc.dtype != torch.float64 or acc.numel() != 7 or cm.dtype != torch.int64 or cm.dim() != 2 or not cm.is_contiguous():
        raise RuntimeError("seg_metrics_accumulate: cm must be int64 [C,C], acc float64[7]")
    if loss is not None and (loss.dtype != torch.float32 or loss.numel() != 1):
        raise RuntimeError("seg_metrics_accumulate: loss must be a float32 scalar tensor")
    call("ur_seg_metrics_accumulate", cm.data_ptr(), cm.shape[0], _p(loss), acc.data_ptr(), _stream())
    return acc


def argmax_channels(logits):
    """preds.argmax(1) for NHWC logits -> int64 [N,H,W] (first maximum wins)."""
    _require_gpu(logits)
    if logits.dim() == 4:
        n, c, h, w = logits.shape
        if not _is_nhwc_dense(logits):
            logits = to_nhwc_raw(logits.contiguous(), logits.dtype)
        out = torch.empty((n, h, w), dtype=torch.int64, device=logits.device)
        rows = n * h * w
    else:
        rows, c = logits.shape
        logits = logits.contiguous()
        out = torch.empty(rows, dtype=torch.int64, device=logits.device)
    call("ur_argmax_channels", logits.data_ptr(), _dt(logits), rows, c, c, out.data_ptr(), _stream())
    return out


def sigmoid_mask(logits):
    """torch.sigmoid(x) > 0.5 -> bool tensor with the logits' shape and layout."""
    _require_gpu(logits)
    out = torch.empty_like(logits, dtype=torch.uint8, memory_format=torch.preserve_format)
    if not (logits.is_contiguous() or (logits.dim() == 4 and _is_nhwc_dense(logits))):
        raise RuntimeError("sigmoid_mask: logits must be dense")
    call("ur_sigmoid_mask", logits.data_ptr(), _dt(logits), logits.numel(), out.data_ptr(), _stream())
    return out.view(torch.bool)
