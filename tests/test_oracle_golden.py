"""Pins the CPU oracle (oracle/uresnet_oracle.py) to vectors recorded from the reference.

The golden files were produced by tests/golden/make_golden.py, which runs the reference's
own model files.  Same torch build => the restatement must reproduce them to float
round-off (thread-count dependent reduction order is the only slack allowed: 2e-6)."""
import hashlib

import numpy as np
import pytest
import torch
import torch.nn as nn

from cases import MODULE_CASES, SEG_MODEL_CASES, OracleNS
from conftest import load_golden, rel_l2, state_from

TOL = 2e-6


def _sha(t):
    return hashlib.sha256(t.detach().contiguous().numpy().tobytes()).hexdigest()


@pytest.mark.parametrize("name", sorted(MODULE_CASES))
def test_block_matches_reference(oracle, name):
    flavour, build = MODULE_CASES[name]
    rec = load_golden(name)
    m = build(OracleNS(oracle, flavour))
    m.load_state_dict(state_from(rec), strict=True)
    m.train()
    xs = []
    i = 0
    while "x%d" % i in rec:
        xs.append(torch.from_numpy(rec["x%d" % i]).requires_grad_(True))
        i += 1
    out = m(*xs)
    assert rel_l2(out, rec["out"]) < TOL
    (out * torch.from_numpy(rec["gout"])).sum().backward()
    for i, x in enumerate(xs):
        assert rel_l2(x.grad, rec["dx%d" % i]) < 20 * TOL, "dx%d" % i
    for k, p in m.named_parameters():
        if "grad/" + k in rec:
            g = torch.from_numpy(rec["grad/" + k])
            assert (p.grad - g).abs().max().item() <= 1e-4 * max(1.0, g.abs().max().item()), k
    sd = m.state_dict()
    for k in rec:
        if k.startswith("w1/"):
            assert rel_l2(sd[k[3:]].double(), rec[k].astype(np.float64)) < TOL, k
    if "out_eval" in rec:
        m.eval()
        with torch.no_grad():
            assert rel_l2(m(*[x.detach() for x in xs]), rec["out_eval"]) < TOL


def test_channel_shuffle_bit_exact(oracle):
    rec = load_golden("channel_shuffle")
    x = torch.from_numpy(rec["x"])
    for g in (2, 4, 8, 16):
        assert torch.equal(oracle.channel_shuffle(x, g), torch.from_numpy(rec["g%d" % g]))
    assert torch.equal(oracle.channel_shuffle(x, 16), x)     # groups == C is the identity


def test_losses_and_masks(oracle):
    rec = load_golden("losses_masks")
    lg = torch.from_numpy(rec["ce_logits"]).requires_grad_(True)
    loss = oracle.compute_loss(lg, torch.from_numpy(rec["ce_target"]))
    loss.backward()
    assert abs(loss.item() - rec["ce_loss"].item()) < 1e-6
    assert rel_l2(lg.grad, rec["ce_dlogits"]) < TOL
    lc = torch.from_numpy(rec["cls_logits"]).requires_grad_(True)
    loss = oracle.cls_loss(lc, torch.from_numpy(rec["cls_target"]))
    loss.backward()
    assert abs(loss.item() - rec["cls_loss"].item()) < 1e-6
    assert rel_l2(lc.grad, rec["cls_dlogits"]) < TOL
    lb = torch.from_numpy(rec["bce_logits"]).requires_grad_(True)
    loss = oracle.bce_loss(lb, torch.from_numpy(rec["bce_target"]))
    loss.backward()
    assert abs(loss.item() - rec["bce_loss"].item()) < 1e-6
    assert rel_l2(lb.grad, rec["bce_dlogits"]) < TOL
    assert torch.equal(oracle.argmax_mask(torch.from_numpy(rec["am_logits"])), torch.from_numpy(rec["am_mask"]))
    assert torch.equal(oracle.sigmoid_mask(torch.from_numpy(rec["sg_logits"])), torch.from_numpy(rec["sg_mask"]))


METRIC_KEYS = ("precision", "recall", "accuracy", "f1", "iou", "dice")


def test_seg_metrics_match_reference(oracle):
    """calculate_metrics against values recorded from the reference's own function (absent labels, zero divisions)."""
    pytest.importorskip("sklearn")
    rec = load_golden("seg_metrics")
    for i in range(3):
        m = oracle.calculate_metrics(torch.from_numpy(rec["logits%d" % i]), torch.from_numpy(rec["target%d" % i]), 19)
        np.testing.assert_allclose([m[k] for k in METRIC_KEYS], rec["metrics%d" % i], rtol=0, atol=1e-15)


@pytest.mark.parametrize("name", sorted(SEG_MODEL_CASES))
def test_seg_model_matches_reference(oracle, name):
    cls, args, classes = SEG_MODEL_CASES[name]
    rec = load_golden(name)
    torch.manual_seed(0)
    m = getattr(oracle, cls)(*args)
    sd = m.state_dict()
    assert list(sd.keys()) == list(rec["keys"])            # state_dict layout is the checkpoint contract
    assert [_sha(v) for v in sd.values()] == list(rec["w0_sha"])   # same seed + same construction order
    m.train()
    x = torch.from_numpy(rec["x"])
    tgt = torch.from_numpy(rec["target"])
    opt = torch.optim.Adam(m.parameters(), lr=1e-4)
    step = oracle.seg_train_step if classes > 1 else oracle.ice_train_step
    losses = []
    for s in range(len(rec["losses"])):
        loss, out = step(m, opt, x, tgt)
        losses.append(loss.item())
        if s == 0:
            assert rel_l2(out, rec["logits"]) < TOL
            grads = dict((k, p.grad) for k, p in m.named_parameters() if p.grad is not None)
            assert list(grads.keys()) == list(rec["grad_keys"])
            for k, n in zip(rec["grad_keys"], rec["grad_norm"]):
                assert abs(grads[k].double().norm().item() - n) <= 1e-3 * n + 1e-6, k
            for k in grads:
                if "grad/" + k in rec:
                    g = torch.from_numpy(rec["grad/" + k])
                    assert (grads[k] - g).abs().max().item() <= 2e-3 * g.abs().max().item() + 1e-6, k
    assert np.allclose(losses, rec["losses"], rtol=1e-5)
    for v, n in zip(m.state_dict().values(), rec["wN_norm"]):
        assert abs(v.double().norm().item() - n) <= 1e-5 * n + 1e-7
    m.eval()
    with torch.no_grad():
        oe = m(x)
    assert rel_l2(oe, rec["logits_eval"]) < 1e-4
    mask = oracle.argmax_mask(oe) if classes > 1 else oracle.sigmoid_mask(oe)
    assert (mask.numpy() != rec["mask_eval"]).mean() < 1e-3


def test_cifar_model_matches_reference(oracle):
    rec = load_golden("cifar_uresnet4_nosu")
    torch.manual_seed(0)
    m = oracle.UResNet4_No_SU(10, 3, scale=3)
    sd = m.state_dict()
    assert list(sd.keys()) == list(rec["keys"])
    assert [_sha(v) for v in sd.values()] == list(rec["w0_sha"])
    m.train()
    x, y = torch.from_numpy(rec["x"]), torch.from_numpy(rec["target"])
    opt = torch.optim.SGD(m.parameters(), lr=0.01)
    losses = []
    for s in range(len(rec["losses"])):
        loss, out = oracle.cls_train_step(m, opt, x, y)
        losses.append(loss.item())
        if s == 0:
            assert rel_l2(out, rec["logits"]) < 1e-5
            nograd = [k for k, p in m.named_parameters() if p.grad is None]
            assert nograd == list(rec["nograd_keys"])       # U_Net.up4.* never receives a gradient
    assert np.allclose(losses, rec["losses"], rtol=1e-4)


TEMP_CASES = {"temp_nlnet": "TempNlNet", "temp_uresnet2": "TempUResNet2", "temp_eresnet": "TempE_ResNet",
              "temp_uresnet": "TempU_ResNet"}

# the remaining classification classes that run as committed: golden -> (oracle class, ctor args, ctor kwargs, SGD lr)
CLS_CASES = {"cifar_uresnet3_noeucb": ("UResNet3_No_EUCB", (10, 3), {"scale": 3}, 0.01),
             "cifar_uresnet3_noublock": ("UResNet3_No_UBlock", (10, 3), {"scale": 3}, 0.01),
             "cifar_eresnet": ("E_ResNet", (), {}, 0.01),
             "mini_uresnet": ("MiniU_ResNet", (100, 3), {}, 0.1)}


@pytest.mark.parametrize("name", sorted(CLS_CASES))
def test_classification_variant_matches_reference(oracle, name):
    """Classification/modules/U_ResNet_CIFAR.py:252,437,487 at 4x3x32x32 and U_ResNet_MiniImageNet.py:321 at 2x3x224x224
    (fixtures made from the reference files): state_dict keys, seeded initialisation, logits, which parameters get no
    gradient, gradient norms and the SGD loss curve."""
    cls, args, kwargs, lr = CLS_CASES[name]
    rec = load_golden(name)
    torch.manual_seed(0)
    m = getattr(oracle, cls)(*args, **kwargs)
    sd = m.state_dict()
    assert list(sd.keys()) == list(rec["keys"])
    assert [_sha(v) for v in sd.values()] == list(rec["w0_sha"])
    m.train()
    x, y = torch.from_numpy(rec["x"]), torch.from_numpy(rec["target"])
    opt = torch.optim.SGD(m.parameters(), lr=lr)
    losses = []
    for s in range(len(rec["losses"])):
        loss, out = oracle.cls_train_step(m, opt, x, y)
        losses.append(loss.item())
        if s == 0:
            assert rel_l2(out, rec["logits"]) < 1e-5
            assert [k for k, p in m.named_parameters() if p.grad is None] == list(rec["nograd_keys"])
            for (k, p), n in zip([(k, p) for k, p in m.named_parameters() if p.grad is not None], rec["grad_norm"]):
                assert abs(p.grad.double().norm().item() - n) <= 1e-3 * n + 1e-6, k
    assert np.allclose(losses, rec["losses"], rtol=1e-4)


def test_dataset_statements_match_reference(oracle):
    """Label remap (Cityscapes.py:191-194, 220-223, 242) and the no-transform image conversion (:238) against outputs of
    those very statements of the reference file.  A.Normalize itself stays unpinned: albumentations is not installed."""
    rec = load_golden("cityscapes_dataset")
    assert tuple(rec["valid_classes"]) == tuple(oracle.CITYSCAPES_VALID_CLASSES) and int(rec["ignore_index"]) == 255
    for i in range(rec["ids"].shape[0]):
        t = oracle.cityscapes_remap(rec["ids"][i])
        assert t.dtype == torch.int64 and np.array_equal(t.numpy(), rec["target"][i])
        img = oracle.normalize_to_tensor(rec["image"][i], mean=None)
        assert img.dtype == torch.float32 and np.array_equal(img.numpy(), rec["image_div255"][i])


@pytest.mark.parametrize("name", sorted(TEMP_CASES))
def test_temp_variant_matches_reference(oracle, name):
    """The scratch variants of Classification/modules/temp.py at 224x224 (fixtures made from the reference file)."""
    rec = load_golden(name)
    torch.manual_seed(0)
    m = getattr(oracle, TEMP_CASES[name])(100, 3)
    sd = m.state_dict()
    assert list(sd.keys()) == list(rec["keys"])
    assert [_sha(v) for v in sd.values()] == list(rec["w0_sha"])
    m.train()
    x, y = torch.from_numpy(rec["x"]), torch.from_numpy(rec["target"])
    opt = torch.optim.SGD(m.parameters(), lr=0.01)
    losses = []
    for s in range(len(rec["losses"])):
        loss, out = oracle.cls_train_step(m, opt, x, y)
        losses.append(loss.item())
        if s == 0:
            assert rel_l2(out, rec["logits"]) < 1e-5
            assert [k for k, p in m.named_parameters() if p.grad is None] == list(rec["nograd_keys"])
            for (k, p), n in zip([(k, p) for k, p in m.named_parameters() if p.grad is not None], rec["grad_norm"]):
                assert abs(p.grad.double().norm().item() - n) <= 1e-3 * n + 1e-6, k
    assert np.allclose(losses, rec["losses"], rtol=1e-4)


@pytest.mark.parametrize("base,shape", [(8, (2, 3, 32, 64)), (32, (1, 3, 36, 52))])
def test_bf16_storage_noise_floor_is_independent_of_any_kernel(oracle, base, shape):
    """How far can two CORRECT implementations of the bf16-storage scheme be apart end to end?  The emulating oracle is
    run twice on UResNetNewEUCB(19,3,base) (S/C:285-346) with identical weights, inputs and rounding points: once with
    fp32 and once with fp64 arithmetic between the rounding points (a 1e-7 relative perturbation of every operator).
    Rounding to bf16 turns such a perturbation d into about sqrt(d * 2^-8) per stored tensor, and 41 BatchNorm layers
    amplify it: the logits end up 2-3e-2 apart, the parameter gradients 0.2 in the median.  That is the floor the GPU
    end-to-end checks are measured against (smoke(), tests/test_hip_fullsize.py); an end-to-end 1e-2 is not attainable by
    any implementation, which is why the 1e-2 bar is enforced per operator and per teacher-forced block instead."""
    torch.manual_seed(0)
    m32 = oracle.UResNetNewEUCB(19, 3, base)
    state = {k: v.clone() for k, v in m32.state_dict().items()}
    m64 = oracle.UResNetNewEUCB(19, 3, base)
    m64.load_state_dict(state)
    m64.double()
    oracle.emulate_bf16_storage(m32)
    oracle.emulate_bf16_storage(m64)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(*shape, generator=g)
    y = torch.randint(0, 19, (shape[0],) + shape[2:], generator=g)
    y[torch.rand(y.shape, generator=g) < 0.1] = 255
    o32 = m32(x)
    l32 = oracle.compute_loss(o32, y)
    l32.backward()
    o64 = m64(x.double())
    l64 = oracle.compute_loss(o64, y)
    l64.backward()
    logits = rel_l2(o32.detach(), o64.detach().float())
    grads = sorted(rel_l2(p.grad, q.grad.float()) for p, q in zip(m32.parameters(), m64.parameters())
                   if p.grad is not None and q.grad.norm() > 0)
    print("\nbf16 noise floor, base %d input %s: logits rel-L2 %.3e, loss %.6f vs %.6f, gradient rel-L2 median %.3e max %.3e"
          % (base, shape, logits, l32.item(), l64.item(), grads[len(grads) // 2], grads[-1]))
    assert 1e-2 < logits < 4e-2              # measured 2.7e-2 / 2.0e-2: above the per-operator bar, by construction of the scheme
    assert abs(l32.item() - l64.item()) < 1e-3 * l64.item()
    assert 5e-2 < grads[len(grads) // 2] < 0.5
