"""ur_ce_fwd_noalloc (the cross-entropy forward whose partial sums live in a device array inside the library, for callers
that cannot allocate) against ur_ce_fwd with a caller-provided workspace: same log-sum-exp rows and the same
{sum, count, mean} bit for bit, on two calls in a row -- the second on a smaller input, so anything the first call left
in the shared array would show -- and a third on a different logits layout.  Sizes beyond the stated limit are rejected."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _inputs(rows, classes, dtype, seed, ignore_frac=0.1):
    g = torch.Generator().manual_seed(seed)
    logits = (torch.randn(rows, classes, generator=g) * 3).to(dtype).to(DEV)
    target = torch.randint(0, classes, (rows,), generator=g)
    target[torch.rand(rows, generator=g) < ignore_frac] = 255
    return logits, target.to(DEV)


def _both(lib_mod, logits, target, classes, ld):
    from uresnet_amd import _lib
    rows = target.numel()
    code = _lib.UR_BF16 if logits.dtype == torch.bfloat16 else _lib.UR_F32
    stream = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(max(int(_lib.query("ur_loss_workspace", rows)), 16), dtype=torch.uint8, device=DEV)
    lse_a = torch.full((rows,), -1.0, dtype=torch.float32, device=DEV)
    out_a = torch.full((3,), -1.0, dtype=torch.float32, device=DEV)
    _lib.call("ur_ce_fwd", logits.data_ptr(), code, target.data_ptr(), rows, classes, ld, 255, lse_a.data_ptr(),
              out_a.data_ptr(), ws.data_ptr(), ws.numel(), stream)
    lse_b = torch.full((rows,), -2.0, dtype=torch.float32, device=DEV)
    out_b = torch.full((3,), -2.0, dtype=torch.float32, device=DEV)
    _lib.call("ur_ce_fwd_noalloc", logits.data_ptr(), code, target.data_ptr(), rows, classes, ld, 255, lse_b.data_ptr(),
              out_b.data_ptr(), stream)
    torch.cuda.synchronize()
    return (lse_a, out_a), (lse_b, out_b)


def test_ce_fwd_noalloc_matches_ce_fwd_on_consecutive_calls():
    import uresnet_amd
    from uresnet_amd import _lib
    _lib.load()
    # first call: a grid at the cap (1024 workgroups); second: a much smaller input (37 workgroups, ragged last one);
    # third: fp32 logits with a pixel pitch wider than the class count (the unstaged kernel)
    for rows, classes, dtype, ld_extra, seed in ((6 * 512 * 1024 // 8, 19, torch.bfloat16, 0, 1),
                                                 (9301, 19, torch.bfloat16, 0, 2),
                                                 (4099, 10, torch.float32, 6, 3)):
        logits, target = _inputs(rows, classes + ld_extra, dtype, seed)
        target = target.clamp(max=255)
        if ld_extra:
            target[(target >= classes) & (target != 255)] = 0
        (lse_a, out_a), (lse_b, out_b) = _both(uresnet_amd, logits, target, classes, classes + ld_extra)
        assert torch.equal(lse_a, lse_b), rows
        assert torch.equal(out_a, out_b), (rows, out_a.tolist(), out_b.tolist())
        assert out_a[1].item() == float((target != 255).sum().item())
        # and against torch on the host, at the tolerance of the other loss tests
        ref = torch.nn.functional.cross_entropy(logits[:, :classes].float().cpu(), target.cpu(), ignore_index=255)
        assert abs(out_b[2].item() - ref.item()) < 1e-5 * max(1.0, abs(ref.item()))


def test_ce_fwd_noalloc_rejects_sizes_beyond_its_limit():
    from uresnet_amd import _lib
    lib = _lib.load()
    limit = _lib.query("ur_ce_fwd_noalloc_max_rows")
    assert limit == 1 << 24
    logits, target = _inputs(64, 19, torch.bfloat16, 4)
    lse = torch.empty(64, dtype=torch.float32, device=DEV)
    out3 = torch.empty(3, dtype=torch.float32, device=DEV)
    rc = lib.ur_ce_fwd_noalloc(logits.data_ptr(), _lib.UR_BF16, target.data_ptr(), limit + 1, 19, 19, 255, lse.data_ptr(),
                               out3.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == -1 and b"exceed the limit" in lib.ur_last_error()


def test_ce_fwd_noalloc_is_capturable_after_prepare():
    """The entry point inside a HIP graph: the per-GPU address of the library's array is resolved by
    ur_ce_fwd_noalloc_prepare() before the capture, the captured call enqueues kernels only, and every replay gives the
    workspace version's bits (the second replay on new logits written into the same buffers)."""
    from uresnet_amd import _lib
    _lib.load()
    _lib.call("ur_ce_fwd_noalloc_prepare")
    rows, classes = 70001, 19
    logits, target = _inputs(rows, classes, torch.bfloat16, 5)
    lse = torch.zeros(rows, dtype=torch.float32, device=DEV)
    out3 = torch.zeros(3, dtype=torch.float32, device=DEV)
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
            _lib.call("ur_ce_fwd_noalloc", logits.data_ptr(), _lib.UR_BF16, target.data_ptr(), rows, classes, classes, 255,
                      lse.data_ptr(), out3.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.current_stream().wait_stream(side)
    ws = torch.empty(max(int(_lib.query("ur_loss_workspace", rows)), 16), dtype=torch.uint8, device=DEV)
    for seed in (6, 7):
        new_logits, new_target = _inputs(rows, classes, torch.bfloat16, seed)
        logits.copy_(new_logits)
        target.copy_(new_target)
        graph.replay()
        lse_ref = torch.empty_like(lse)
        out_ref = torch.empty_like(out3)
        _lib.call("ur_ce_fwd", logits.data_ptr(), _lib.UR_BF16, target.data_ptr(), rows, classes, classes, 255,
                  lse_ref.data_ptr(), out_ref.data_ptr(), ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert torch.equal(lse, lse_ref) and torch.equal(out3, out_ref), seed
