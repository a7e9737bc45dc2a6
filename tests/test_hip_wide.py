"""GPU parity of the 128-channel 3x3 kernel (ur_conv3x3_wide_*), called through the C ABI.

Reference: torch fp32 convolutions of the same bf16-rounded operands (tolerance 1e-2 relative L2 on bf16 results, the
per-operator bar of test_hip_parity.py; the fp32 statistics and BatchNorm sums to 1e-4), and the tile kernel on the same
operands (two fp32 sums in different orders, then one bf16 rounding: 2e-3).  The shapes cover a single tile, tiles
ragged in both directions, two channel blocks, a one-slice reduction (Cin = 32), more items than workgroups (a
workgroup walks several tiles), and tensors that are channel windows of wider ones (pitches)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# n, h, w, cin, cout
SHAPES = [
    (1, 16, 32, 32, 128),      # one tile, one slice
    (2, 37, 50, 64, 128),      # ragged rows and columns
    (1, 20, 70, 128, 256),     # two channel blocks, three tile columns
    (3, 33, 65, 96, 128),      # three slices, one extra column of tiles for a single pixel column
    (5, 128, 256, 32, 128),    # 320 items on 256 workgroups: the persistent loop takes a second item
]


@pytest.fixture(scope="module")
def lib():
    import uresnet_amd
    uresnet_amd._lib.load()
    return uresnet_amd._lib


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def desc(lib, n, h, w, cin, cout, in_ld=0, out_ld=0):
    return lib.ConvDesc(lib.UR_BF16, n, h, w, cin, h, w, cout, 3, 3, 1, 1, in_ld or cin, out_ld or cout)


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def operands(shape, seed):
    n, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, cin, h, w, generator=g).relu_().to(torch.bfloat16)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    dy = torch.randn(n, cout, h, w, generator=g).to(torch.bfloat16)
    return x, wt, dy


def packed(lib, kern, wt, d, dgrad):
    img = torch.empty(lib.query("ur_conv3x3_%s_packed_elems" % kern, C.byref(d)), dtype=torch.bfloat16, device=DEV)
    lib.call("ur_conv3x3_%s_pack_weight" % kern, wt.data_ptr(), img.data_ptr(), C.byref(d), dgrad, stream())
    return img


@pytest.mark.parametrize("shape", SHAPES)
def test_wide_forward_and_statistics(lib, shape):
    n, h, w, cin, cout = shape
    x, wt, _ = operands(shape, 11)
    ref = F.conv2d(x.float(), wt.to(torch.bfloat16).float(), padding=1)
    xd, wd = nhwc(x).to(DEV), wt.to(DEV)
    d = desc(lib, n, h, w, cin, cout)
    assert lib.query("ur_conv3x3_wide_supported", C.byref(d), 0) == 1
    out = {}
    for kern in ("wide", "tile"):
        y = torch.full((n, h, w, cout), float("nan"), dtype=torch.bfloat16, device=DEV)
        rows = lib.query("ur_conv3x3_%s_stats_rows" % kern, C.byref(d))
        st = torch.full((rows, cout, 2), float("nan"), device=DEV)
        lib.call("ur_conv3x3_%s_fwd" % kern, xd.data_ptr(), packed(lib, kern, wd, d, 0).data_ptr(), y.data_ptr(), st.data_ptr(),
                 C.byref(d), stream())
        out[kern] = (y.float().permute(0, 3, 1, 2).cpu(), st.sum(0).cpu())
    assert rel_l2(out["wide"][0], ref) < 1e-2
    assert rel_l2(out["wide"][0], out["tile"][0]) < 2e-3
    rd = ref.double()
    assert rel_l2(out["wide"][1][:, 0], rd.sum((0, 2, 3))) < 1e-4
    assert rel_l2(out["wide"][1][:, 1], (rd * rd).sum((0, 2, 3))) < 1e-4


@pytest.mark.parametrize("shape", SHAPES[:4])
def test_wide_backward_data_variants(lib, shape):
    """Plain, + addend, and with the BatchNorm-backward sums of the tensor the gradient belongs to."""
    n, h, w, cin, cout = shape
    if cin % 128:
        cin = 128                                   # backward-data produces the layer's input channels: a multiple of 128
    shape = (n, h, w, cin, cout)
    _, wt, dy = operands(shape, 12)
    g = torch.Generator().manual_seed(13)
    ref = torch.nn.grad.conv2d_input((n, cin, h, w), wt.to(torch.bfloat16).float(), dy.float(), padding=1)
    add = torch.randn(n, cin, h, w, generator=g).to(torch.bfloat16)
    bny = torch.randn(n, cin, h, w, generator=g).to(torch.bfloat16)
    mean, inv = torch.randn(cin, generator=g) * 0.1, torch.rand(cin, generator=g) + 0.5
    gam, bet = torch.randn(cin, generator=g), torch.randn(cin, generator=g) * 0.3
    d = desc(lib, n, h, w, cin, cout)
    assert lib.query("ur_conv3x3_wide_supported", C.byref(d), 1) == 1
    dyd, wd = nhwc(dy).to(DEV), wt.to(DEV)
    img = packed(lib, "wide", wd, d, 1)
    dx = [torch.full((n, h, w, cin), float("nan"), dtype=torch.bfloat16, device=DEV) for _ in range(3)]
    lib.call("ur_conv3x3_wide_dgrad", dyd.data_ptr(), img.data_ptr(), dx[0].data_ptr(), C.byref(d), stream())
    addd = nhwc(add).to(DEV)
    lib.call("ur_conv3x3_wide_dgrad_add", dyd.data_ptr(), img.data_ptr(), dx[1].data_ptr(), addd.data_ptr(), cin, C.byref(d), stream())
    rows = lib.query("ur_conv3x3_wide_stats_rows", C.byref(d))
    part = torch.full((rows, cin, 2), float("nan"), device=DEV)
    bnyd = nhwc(bny).to(DEV)
    consts = [t.to(DEV) for t in (mean, inv, gam, bet)]
    lib.call("ur_conv3x3_wide_dgrad_bnsums", dyd.data_ptr(), img.data_ptr(), dx[2].data_ptr(), bnyd.data_ptr(), cin,
             consts[0].data_ptr(), consts[1].data_ptr(), consts[2].data_ptr(), consts[3].data_ptr(), part.data_ptr(), C.byref(d),
             stream())
    got = [t.float().permute(0, 3, 1, 2).cpu() for t in dx]
    assert rel_l2(got[0], ref) < 1e-2
    assert torch.equal(got[2], got[0])              # the sums are a by-product: the gradient itself is the plain one
    # addend: the rounded gradient plus the addend, rounded again (what a separate add of two bf16 tensors gives)
    assert torch.equal(got[1], (got[0] + add.float()).to(torch.bfloat16).float())
    # sums over the stored (rounded) gradient where the BatchNorm's ReLU passed: s1 = sum g, s2 = sum g * x_hat
    c = lambda t: t.view(1, -1, 1, 1)  # noqa: E731
    xh = (bny.float() - c(mean)) * c(inv)
    gmask = torch.where(xh * c(gam) + c(bet) > 0, got[0], torch.zeros(()))
    sums = part.sum(0).cpu()
    assert rel_l2(sums[:, 0], gmask.double().sum((0, 2, 3))) < 1e-4
    assert rel_l2(sums[:, 1], (gmask.double() * xh.double()).sum((0, 2, 3))) < 1e-4


def test_wide_reads_and_writes_channel_windows(lib):
    """Input and output as channel windows of wider tensors (the copy-free concatenation of the decoder): pitches in the
    descriptor, channels outside the windows untouched."""
    n, h, w, cin, cout = 2, 19, 40, 64, 128
    x, wt, _ = operands((n, h, w, cin, cout), 14)
    ref = F.conv2d(x.float(), wt.to(torch.bfloat16).float(), padding=1)
    xin = torch.full((n, h, w, cin + 32), 7.0, dtype=torch.bfloat16, device=DEV)
    xin[..., 16:16 + cin] = nhwc(x).to(DEV)
    yout = torch.full((n, h, w, cout + 64), -3.0, dtype=torch.bfloat16, device=DEV)
    d = desc(lib, n, h, w, cin, cout, in_ld=cin + 32, out_ld=cout + 64)
    img = packed(lib, "wide", wt.to(DEV), d, 0)
    lib.call("ur_conv3x3_wide_fwd", xin[..., 16:].data_ptr(), img.data_ptr(), yout[..., 24:].data_ptr(), None, C.byref(d), stream())
    got = yout[..., 24:24 + cout].float().permute(0, 3, 1, 2).cpu()
    assert rel_l2(got, ref) < 1e-2
    assert bool((yout[..., :24] == -3.0).all()) and bool((yout[..., 24 + cout:] == -3.0).all())


def test_wide_image_from_the_batched_packer(lib):
    """A row of ur_conv3x3_tile_pack_weights_multi with bit 1 of its direction word set builds the same image as
    ur_conv3x3_wide_pack_weight (the repack of every layer after an optimiser step goes through the batched launch)."""
    cin, cout = 64, 256
    g = torch.Generator().manual_seed(15)
    wt = torch.randn(cout, cin, 3, 3, generator=g).to(DEV)
    d = desc(lib, 1, 16, 32, cin, cout)
    rows, first = [], 0
    imgs = {}
    for dgrad in (0, 1):
        if not lib.query("ur_conv3x3_wide_supported", C.byref(d), dgrad):
            continue
        imgs[dgrad] = (packed(lib, "wide", wt, d, dgrad), torch.zeros(cin * cout * 9, dtype=torch.bfloat16, device=DEV))
        rows.append((wt.data_ptr(), imgs[dgrad][1].data_ptr(), cin, cout, dgrad | 2, first))
        first += lib.query("ur_conv3x3_tile_pack_blocks", cin, cout, dgrad)
    assert imgs
    table = torch.tensor(rows, dtype=torch.int64).to(DEV)
    lib.call("ur_conv3x3_tile_pack_weights_multi", table.data_ptr(), len(rows), first, stream())
    for single, multi in imgs.values():
        assert torch.equal(single.view(torch.int16), multi.view(torch.int16))


def test_wide_rejects_what_it_cannot_do(lib):
    d = desc(lib, 1, 16, 32, 64, 64)               # 64 produced channels: not a multiple of 128
    assert lib.query("ur_conv3x3_wide_supported", C.byref(d), 0) == 0 and lib.query("ur_conv3x3_wide_preferred", C.byref(d), 0) == 0
    buf = torch.zeros(64 * 64 * 9, dtype=torch.bfloat16, device=DEV)
    rc = lib.load().ur_conv3x3_wide_fwd(buf.data_ptr(), buf.data_ptr(), buf.data_ptr(), None, C.byref(d), stream())
    assert rc != 0 and b"128" in lib.load().ur_last_error()
    d = desc(lib, 1, 16, 32, 48, 128)              # reduction not a multiple of 32
    assert lib.query("ur_conv3x3_wide_supported", C.byref(d), 0) == 0
    # small problems stay with the tile kernel even where the shape is supported
    d = desc(lib, 1, 16, 32, 32, 128)
    assert lib.query("ur_conv3x3_wide_supported", C.byref(d), 0) == 1 and lib.query("ur_conv3x3_wide_preferred", C.byref(d), 0) == 0


@pytest.mark.parametrize("shape", [(2, 37, 50, 64, 128), (1, 20, 70, 128, 64)])
def test_backward_weight_with_a_marker_event_between_its_two_kernels(lib, shape):
    """ur_wgrad3x3_tile_marked: the same two kernels as ur_wgrad3x3_tile (bit-identical gradient, checked against torch's
    fp32 gradient of the same bf16 operands), with the caller's event recorded between them: start -> marker -> end are
    ordered on the stream and both intervals are measurable.  The 32 x 32-block kernel refuses an event."""
    n, h, w, cin, cout = shape
    x, wt, dy = operands(shape, 21)
    xd, dyd = nhwc(x).to(DEV), nhwc(dy).to(DEV)
    d = desc(lib, n, h, w, cin, cout)
    assert lib.query("ur_wgrad3x3_tile_supported", C.byref(d)) == 1
    nws = lib.query("ur_wgrad3x3_tile_workspace", C.byref(d))
    ws = torch.empty(nws, dtype=torch.uint8, device=DEV)
    dw_a = torch.empty(cout, cin, 3, 3, device=DEV)
    dw_b = torch.full_like(dw_a, 7.0)
    lib.call("ur_wgrad3x3_tile", xd.data_ptr(), dyd.data_ptr(), dw_a.data_ptr(), ws.data_ptr(), nws, C.byref(d), stream())
    st = torch.cuda.current_stream()
    e0, mid, e1 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    mid.record(st)
    e0.record(st)
    lib.call("ur_wgrad3x3_tile_marked", xd.data_ptr(), dyd.data_ptr(), dw_b.data_ptr(), ws.data_ptr(), nws, C.byref(d),
             C.c_void_p(mid.cuda_event), stream())
    e1.record(st)
    torch.cuda.synchronize()
    assert torch.equal(dw_a, dw_b)
    assert e0.elapsed_time(mid) > 0 and mid.elapsed_time(e1) > 0
    ref = torch.nn.grad.conv2d_weight(x.float(), wt.shape, dy.float(), padding=1)
    assert rel_l2(dw_b.cpu(), ref) < 1e-4
    d32 = desc(lib, 1, 16, 32, 32, 32)
    if lib.query("ur_wgrad3x3_tile_supported", C.byref(d32)) == 2:
        buf = torch.zeros(1 << 20, dtype=torch.uint8, device=DEV)
        with pytest.raises(lib.UResNetHipError, match="takes no event"):
            lib.call("ur_wgrad3x3_tile_marked", buf.data_ptr(), buf.data_ptr(), buf.data_ptr(), buf.data_ptr(), buf.numel(),
                     C.byref(d32), C.c_void_p(mid.cuda_event), stream())
