"""GPU parity at the headline benchmark size: one Cityscapes-shaped 512x1024 image through
UResNetNewEUCB(19,3,64) (config 3), one 224x224 MiniImageNet batch through U_ResNet(100,3) (config 2)
and the Ice-Seg inference path (config 5), each against the CPU oracle run on the same seeded inputs."""
import numpy as np
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def pkg():
    import uresnet_amd
    uresnet_amd._lib.load()
    return uresnet_amd


def host(pkg, t):
    t = t.detach()
    if t.dim() == 4:
        return pkg.ops.to_nchw_raw(pkg.ops._g(t), torch.float32).cpu()
    return t.float().cpu()


def _seg_inputs(h, w, classes=19, b=1):
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(b, 3, h, w, generator=g)
    y = torch.randint(0, classes, (b, h, w), generator=g)
    y[torch.rand(y.shape, generator=g) < 0.1] = 255
    return x, y


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_config3_full_resolution_step_matches_oracle(pkg, oracle, dtype):
    """fp32 storage is checked against the plain oracle; bf16 storage against the oracle with bf16 storage
    emulation (same tensors rounded, fp32 arithmetic inside), which keeps ReLU masks comparable."""
    torch.manual_seed(0)
    ref = oracle.UResNetNewEUCB(19, 3, 64).train()
    bf16 = dtype == torch.bfloat16
    if bf16:
        oracle.emulate_bf16_storage(ref)
    state0 = {k: v.clone() for k, v in ref.state_dict().items()}
    x, y = _seg_inputs(512, 1024)
    ro = ref(x)
    rl = oracle.compute_loss(ro, y)
    rl.backward()
    m = pkg.modules.U_ResNet.UResNetNewEUCB(19, 3, 64).set_compute_dtype(dtype)
    m.load_state_dict(state0)
    m.to(DEV).train()
    out = m(x.to(DEV))
    loss = pkg.train.compute_loss(out, y.to(DEV))
    loss.backward()
    assert out.shape == (1, 19, 512, 1024)
    # bf16: two correct implementations of the storage scheme are 2-2.7e-2 apart on logits (fp32 vs fp64 arithmetic in
    # the emulating oracle, tests/test_oracle_golden.py::test_bf16_storage_noise_floor_...); measured here 1.6e-2
    assert rel_l2(host(pkg, out), ro.detach()) < (2e-4 if not bf16 else 2.5e-2)
    assert abs(loss.item() - rl.item()) < (1e-5 if not bf16 else 2e-3) * abs(rl.item())
    # Gradients: with 5e5 pixels x 41 ReLU layers a handful of pre-activations sit within round-off of zero and
    # flip their mask (each flip shifts every upstream gradient, see tests/test_hip_parity.py); the reference's
    # own fp32-vs-fp64 gradients differ by 1e-3 median at 64x128.
    gref = dict((k, p.grad) for k, p in ref.named_parameters())
    gmax = max(g.norm() for g in gref.values())
    errs = []
    for k, p in m.named_parameters():
        if p.grad is None or gref[k].norm() < 1e-6 * gmax:
            continue
        errs.append(rel_l2(p.grad.cpu(), gref[k]))
    # bf16: the randomly initialised 41-BN-layer network amplifies a 1-ulp bf16 difference in one activation by
    # ~1e4 on the way back (measured: two bf16 runs that differ only in fp32 summation order -- tiled vs generic
    # kernels -- disagree by 3e-2 at the first layer reached in backward and 2.5e-1 at the deepest); the reference's
    # own bf16-autocast gradients sit 0.3-0.46 from its fp32 ones.  Per-operator tests pin the kernels at 1e-2.
    print("\nconfig 3 chained %s: logits rel-L2 %.2e, loss %.6f vs %.6f, parameter-gradient rel-L2 median %.2e max %.2e"
          % (dtype, rel_l2(host(pkg, out), ro.detach()), loss.item(), rl.item(), np.median(errs), max(errs)))
    assert np.median(errs) < (2e-2 if not bf16 else 0.6), np.median(errs)
    # running statistics updated like PyTorch (momentum 0.1, unbiased variance)
    sd_r, sd_m = ref.state_dict(), m.state_dict()
    for k in ("conv1.1.running_mean", "up4.conv.4.running_var", "out_conv.0.running_var", "Upsample.up_dwc2.1.running_mean"):
        assert rel_l2(sd_m[k].cpu(), sd_r[k]) < (1e-4 if not bf16 else 1e-2), k
    assert int(sd_m["conv1.1.num_batches_tracked"]) == 1


def test_config3_tiled_and_generic_kernels_agree_at_full_size(pkg):
    """Two independent HIP implementations of the 3x3 layers (LDS-tiled vs gather implicit GEMM) must produce the
    same bf16 training step on a full-resolution image."""
    torch.manual_seed(0)
    x, y = _seg_inputs(512, 1024)
    res = {}
    for tile in (True, False):
        pkg.ops.use_tile_kernel = tile
        try:
            torch.manual_seed(0)
            m = pkg.modules.U_ResNet.UResNetNewEUCB(19, 3, 64).set_compute_dtype(torch.bfloat16).to(DEV).train()
            out = m(x.to(DEV))
            loss = pkg.train.compute_loss(out, y.to(DEV))
            loss.backward()
            res[tile] = (host(pkg, out), loss.item(), m.in_conv[3].weight.grad.cpu(), m.down4[1][3].weight.grad.cpu())
        finally:
            pkg.ops.use_tile_kernel = True
    assert rel_l2(res[True][0], res[False][0]) < 2e-2
    assert abs(res[True][1] - res[False][1]) < 2e-3
    # gradients: the two kernels round y differently in the last bf16 bit now and then, which flips ReLU masks
    assert rel_l2(res[True][2], res[False][2]) < 0.3
    assert rel_l2(res[True][3], res[False][3]) < 0.3


def test_config3_step_is_bit_reproducible_run_to_run(pkg):
    """Split reductions (backward-weight slabs, BatchNorm partial rows, loss partials) are summed in a fixed order and no
    kernel on the path uses floating-point atomics: two training steps from the same state on the same 2 x 512 x 1024
    batch give bit-identical logits, loss, gradients and updated weights -- with the tile and the 128-channel kernels
    both in play (the second image pair makes the 128- and 256-channel levels large enough for the latter)."""
    x, y = _seg_inputs(512, 1024, b=2)
    runs = []
    for _ in range(2):
        torch.manual_seed(0)
        m = pkg.modules.U_ResNet.UResNetNewEUCB(19, 3, 64).set_compute_dtype(torch.bfloat16).to(DEV).train()
        opt = pkg.train.Adam(m.parameters(), lr=1e-4)
        pkg.ops.kernel_timings = []
        try:
            loss, out = pkg.train.seg_train_step(m, opt, x.to(DEV), y.to(DEV))
        finally:
            tags, pkg.ops.kernel_timings = set(t[0] for t in pkg.ops.kernel_timings), None
        assert {"tile_fwd", "wide_fwd", "tile_dgrad_bn", "wide_dgrad_bn", "tile_wgrad"} <= tags
        runs.append((out.detach().clone(), loss.detach().clone(), [p.grad.clone() for p in m.parameters() if p.grad is not None],
                     [p.detach().clone() for p in m.parameters()]))
    a, b = runs
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert len(a[2]) == len(b[2]) and all(torch.equal(u, v) for u, v in zip(a[2], b[2]))
    assert all(torch.equal(u, v) for u, v in zip(a[3], b[3]))


def test_config2_mini_imagenet_step_matches_oracle(pkg, oracle):
    torch.manual_seed(0)
    ref = oracle.MiniU_ResNet(100, 3).train()
    g = torch.Generator().manual_seed(7)
    x = torch.randn(2, 3, 224, 224, generator=g)
    y = torch.randint(0, 100, (2,), generator=g)
    ro = ref(x)
    rl = oracle.cls_loss(ro, y)
    rl.backward()
    m = pkg.modules.U_ResNet_MiniImageNet.U_ResNet(100, 3).set_compute_dtype(torch.float32)
    m.load_state_dict(ref.state_dict())
    m.to(DEV).train()
    out = m(x.to(DEV))
    loss = pkg.train.cross_entropy_loss(out, y.to(DEV))
    loss.backward()
    assert rel_l2(out.float().cpu(), ro.detach()) < 5e-4
    assert abs(loss.item() - rl.item()) < 1e-4
    unused_ref = sorted(k for k, p in ref.named_parameters() if p.grad is None)
    unused = sorted(k for k, p in m.named_parameters() if p.grad is None)
    assert unused == unused_ref and len(unused) > 0         # up2..up4 and CBPooling never receive gradients
    assert rel_l2(m.fc.weight.grad.cpu(), ref.fc.weight.grad) < 2e-3


def test_config5_ice_seg_inference_mask(pkg, oracle):
    """eval(): running-statistics BN, sigmoid(logits) > 0.5 mask; batch of 4 512x512 images."""
    torch.manual_seed(0)
    ref = oracle.UResNet(1, 3, 64)
    # a few training steps on small crops so the running statistics are not the initial (0, 1)
    g = torch.Generator().manual_seed(3)
    ref.train()
    with torch.no_grad():
        for _ in range(2):
            ref(torch.randn(2, 3, 64, 64, generator=g))
    ref.eval()
    x = torch.randn(4, 3, 512, 512, generator=g)
    with torch.no_grad():
        ro = ref(x)
    want = oracle.sigmoid_mask(ro)
    for dtype, tol, flips in ((torch.float32, 2e-4, 1e-4), (torch.bfloat16, 6e-2, 3e-2)):
        m = pkg.modules.U_ResNet.UResNet(1, 3, 64).set_compute_dtype(dtype)
        m.load_state_dict(ref.state_dict())
        m.to(DEV).eval()
        with torch.no_grad():
            out = m(x.to(DEV))
            mask = pkg.train.sigmoid_mask(out)
        assert rel_l2(host(pkg, out), ro) < tol
        got = host(pkg, mask.float()).bool()
        assert (got != want).float().mean().item() < flips       # only logits within round-off of 0 may flip


def test_config5_large_batch_inference_equals_small_batches_bit_exactly(pkg):
    """Config 5 at its large batch (32 x 3 x 512 x 512, the size tools/bench_eval.py times; the CPU oracle would take
    minutes): eval-mode BatchNorm uses fixed statistics, so every image is independent of its batch -- the logits and
    masks of one batch of 32 equal those of the same images sent through in batches of 4 bit for bit (tile order, XCD
    mapping, folded-BatchNorm epilogues and the two-stream forward do not depend on what else is in the batch), and the
    masks agree with the one-stream forward."""
    ops = pkg.ops
    torch.manual_seed(0)
    m = pkg.modules.U_ResNet.UResNet(1, 3, 64).set_compute_dtype(torch.bfloat16).to(DEV)
    g = torch.Generator().manual_seed(5)
    m.train()
    with torch.no_grad():
        for _ in range(2):                                  # running statistics away from their initial (0, 1)
            m(torch.randn(2, 3, 64, 64, generator=g).to(DEV))
    m.eval()
    x = torch.randn(32, 3, 512, 512, generator=g).to(DEV)
    with torch.no_grad():
        big = m(x)
        small = torch.cat([m(x[i:i + 4]) for i in range(0, 32, 4)])
        old, ops.branch_streams = ops.branch_streams, not ops.branch_streams
        try:
            other = m(x)
        finally:
            ops.branch_streams = old
    assert torch.equal(big, small)
    assert torch.equal(big, other)
    assert torch.isfinite(big.float()).all()
    assert torch.equal(pkg.train.sigmoid_mask(big), pkg.train.sigmoid_mask(small))


def test_config4_full_resolution_1024x2048(pkg, oracle):
    """Config 4: one 1024x2048 image through UResNetNewEUCB(19,3,64).  Forward logits and loss against the oracle (train
    mode, batch statistics; fp32 2e-4, bf16 against the bf16-storage oracle), then properties that need no oracle: the
    LDS-tiled and the generic kernels agree on a whole bf16 training step, running statistics follow the oracle's, two
    Adam steps keep the loss finite and falling, and a row too long for the tile kernel's 24-bit offsets is refused by
    its support query (the generic kernel takes such a layer)."""
    import ctypes as C
    from uresnet_amd import _lib
    x, y = _seg_inputs(1024, 2048)
    torch.manual_seed(0)
    ref = oracle.UResNetNewEUCB(19, 3, 64).train()
    state0 = {k: v.clone() for k, v in ref.state_dict().items()}
    with torch.no_grad():
        ro = ref(x)
        rl = oracle.compute_loss(ro, y)
    stats_ref = {k: v.clone() for k, v in ref.state_dict().items() if "running" in k}
    ref.load_state_dict(state0)
    hb = oracle.emulate_bf16_storage(ref)
    with torch.no_grad():
        ro_b = ref(x)
        rl_b = oracle.compute_loss(ro_b, y)
    del ref, hb
    xd, yd = x.to(DEV), y.to(DEV)
    res = {}
    for dtype, tile in ((torch.float32, True), (torch.bfloat16, True), (torch.bfloat16, False)):
        pkg.ops.use_tile_kernel = tile
        try:
            m = pkg.modules.U_ResNet.UResNetNewEUCB(19, 3, 64).set_compute_dtype(dtype)
            m.load_state_dict(state0)
            m.to(DEV).train()
            opt = pkg.train.Adam(m.parameters(), lr=1e-4)
            loss, out = pkg.train.seg_train_step(m, opt, xd, yd)
            assert out.shape == (1, 19, 1024, 2048)
            logits = host(pkg, out)
            if dtype == torch.float32:
                assert rel_l2(logits, ro) < 2e-4
                assert abs(loss.item() - rl.item()) < 1e-5 * abs(rl.item())
                for k, v in stats_ref.items():
                    assert rel_l2(m.state_dict()[k].cpu(), v) < 1e-4, k
            else:
                assert rel_l2(logits, ro_b) < 3e-2      # the scheme's own noise floor is 2-2.7e-2 (see the oracle test)
                assert abs(loss.item() - rl_b.item()) < 2e-3 * abs(rl_b.item())
            loss2, _ = pkg.train.seg_train_step(m, opt, xd, yd)
            assert torch.isfinite(loss2).item() and loss2.item() < loss.item()
            res[(dtype, tile)] = (logits, loss.item(), loss2.item())
            del m, opt, out
        finally:
            pkg.ops.use_tile_kernel = True
    a, b = res[(torch.bfloat16, True)], res[(torch.bfloat16, False)]
    assert rel_l2(a[0], b[0]) < 2e-2 and abs(a[1] - b[1]) < 2e-3 and abs(a[2] - b[2]) < 2e-2
    # the tile kernels address a pixel row with 24-bit element offsets: w * pitch must stay below 2^24
    d = _lib.ConvDesc()
    d.dtype, d.n, d.h, d.w, d.cin, d.oh, d.ow, d.cout = _lib.UR_BF16, 1, 32, 2048, 64, 32, 2048, 64
    d.kh = d.kw = 3; d.stride = d.pad = 1; d.in_ld = d.out_ld = 64
    assert _lib.query("ur_conv3x3_tile_supported", C.byref(d), 0) == 1
    d.in_ld = 8192                                   # a 64-channel window of an 8192-channel tensor: 2048 * 8192 = 2^24
    assert _lib.query("ur_conv3x3_tile_supported", C.byref(d), 0) == 0


def test_config2_bf16_training_step_at_batch_32(pkg, oracle):
    """Config 2 as stated: U_ResNet(100, 3) (U_ResNet_MiniImageNet.py:321) on a 32 x 3 x 224 x 224 batch, bf16, SGD lr 0.1
    (CNN-MiniImageNet.py:36-51, 84-94).  Logits and loss of the first step against the oracle with bf16 storage emulation;
    the second step's loss must be finite (lr 0.1 on random data may go up or down)."""
    torch.manual_seed(0)
    ref = oracle.MiniU_ResNet(100, 3).train()
    oracle.emulate_bf16_storage(ref)
    state0 = {k: v.clone() for k, v in ref.state_dict().items()}
    g = torch.Generator().manual_seed(7)
    x = torch.randn(32, 3, 224, 224, generator=g)
    y = torch.randint(0, 100, (32,), generator=g)
    with torch.no_grad():
        ro = ref(x)
        rl = oracle.cls_loss(ro, y)
    m = pkg.modules.U_ResNet_MiniImageNet.U_ResNet(100, 3).set_compute_dtype(torch.bfloat16)
    m.load_state_dict(state0)
    m.to(DEV).train()
    opt = pkg.train.SGD(m.parameters(), lr=0.1)
    loss, out = pkg.train.cls_train_step(m, opt, x.to(DEV), y.to(DEV))
    print("\nconfig 2 bf16 B=32: logits rel-L2 %.2e, loss %.5f vs %.5f" % (rel_l2(out.float().cpu(), ro), loss.item(), rl.item()))
    assert rel_l2(out.float().cpu(), ro) < 3e-2
    assert abs(loss.item() - rl.item()) < 1e-2 * abs(rl.item())
    loss2, _ = pkg.train.cls_train_step(m, opt, x.to(DEV), y.to(DEV))
    assert torch.isfinite(loss2).item()
