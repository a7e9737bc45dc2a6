"""CPU-side checks of the drop-in boundary: the shared library loads, exports every symbol the
header declares, the ctypes table covers the header, and the model classes keep the
reference's constructor signatures and state_dict layout.  No kernel is launched here."""
import ctypes
import hashlib
import inspect
import re

import pytest
import torch

import uresnet_amd
from uresnet_amd import _lib
from conftest import load_golden


def _header_symbols():
    text = open(_lib.HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ur_[a-zA-Z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = _header_symbols()
    assert len(syms) >= 45
    for s in syms:
        assert hasattr(lib, s), "liburesnet_hip.so does not export %s" % s


def test_ctypes_table_matches_header():
    assert sorted(_lib.SIGNATURES) == _header_symbols()
    lib = _lib.load()
    assert lib.ur_version() >= 100
    assert lib.ur_last_error() is not None


def test_bad_arguments_return_error_not_crash():
    d = _lib.ConvDesc()
    with pytest.raises(_lib.UResNetHipError, match="bad dtype|non-positive|null"):
        _lib.call("ur_conv2d_fwd", None, None, None, None, None, ctypes.byref(d), None)
    assert _lib.query("ur_conv2d_packed_elems", None, 0) == 0


def test_cpu_tensors_are_rejected_loudly():
    from uresnet_amd.modules import U_ResNet as M
    blk = M.DoubleConv(3, 8)
    with pytest.raises(_lib.UResNetHipError, match="no CPU fallback"):
        blk(torch.randn(1, 3, 8, 8))


@pytest.mark.parametrize("name,cls,args", [
    ("seg_newEUCB_c8", "UResNetNewEUCB", (19, 3, 8)),
    ("seg_uresnet_c8_bce", "UResNet", (1, 3, 8)),
    ("seg_noEUCB_c8", "UResNetNoEUCB", (19, 3, 8)),
    ("ice_noresblock_c8", "UResNetNoResBlock", (1, 3, 8)),
])
def test_seg_state_dict_layout_and_seeded_init(name, cls, args):
    import hashlib
    from uresnet_amd.modules import U_ResNet as M
    rec = load_golden(name)
    torch.manual_seed(0)
    m = getattr(M, cls)(*args)
    sd = m.state_dict()
    assert list(sd.keys()) == list(rec["keys"])
    sha = [hashlib.sha256(v.contiguous().numpy().tobytes()).hexdigest() for v in sd.values()]
    assert sha == list(rec["w0_sha"])          # same construction order => same default initialisation


def test_cifar_state_dict_layout():
    import hashlib
    from uresnet_amd.modules import U_ResNet_CIFAR as M
    rec = load_golden("cifar_uresnet4_nosu")
    torch.manual_seed(0)
    m = M.UResNet4_No_SU(10, 3, scale=3)
    sd = m.state_dict()
    assert list(sd.keys()) == list(rec["keys"])
    sha = [hashlib.sha256(v.contiguous().numpy().tobytes()).hexdigest() for v in sd.values()]
    assert sha == list(rec["w0_sha"])


@pytest.mark.parametrize("name, cls", [("temp_nlnet", "NlNet"), ("temp_uresnet2", "UResNet2"),
                                       ("temp_eresnet", "E_ResNet"), ("temp_uresnet", "U_ResNet")])
def test_temp_variant_state_dict_layout(name, cls):
    import hashlib
    from uresnet_amd.modules import temp as M
    rec = load_golden(name)
    torch.manual_seed(0)
    sd = getattr(M, cls)(100, 3).state_dict()
    assert list(sd.keys()) == list(rec["keys"])
    sha = [hashlib.sha256(v.contiguous().numpy().tobytes()).hexdigest() for v in sd.values()]
    assert sha == list(rec["w0_sha"])


def test_shape_broken_cifar_classes_keep_signature_and_parameters():
    """U_ResNet / S_ResNet / NlNet of U_ResNet_CIFAR.py (:296, :342, :389) cannot run at 32x32 in the reference; the
    constructors still exist with the reference's signature, state_dict keys and seeded initialisation."""
    from uresnet_amd.modules import U_ResNet_CIFAR as Cc
    rec = load_golden("cifar_broken_classes")
    sig = lambda c: list(inspect.signature(c.__init__).parameters)[1:]
    assert sig(Cc.U_ResNet) == sig(Cc.S_ResNet) == ["num_classes", "in_channels", "ResidulBlock"]
    assert sig(Cc.NlNet) == ["num_classes", "in_channels", "ResidulBlock", "scale"]
    for name, build in (("U_ResNet", lambda: Cc.U_ResNet()), ("S_ResNet", lambda: Cc.S_ResNet()),
                        ("NlNet", lambda: Cc.NlNet(10, 3))):
        torch.manual_seed(0)
        sd = build().state_dict()
        assert list(sd.keys()) == list(rec[name + "/keys"]), name
        sha = [hashlib.sha256(v.contiguous().numpy().tobytes()).hexdigest() for v in sd.values()]
        assert sha == list(rec[name + "/w0_sha"]), name


def test_constructor_signatures_follow_the_reference():
    from uresnet_amd.modules import U_ResNet as S, U_ResNet_CIFAR as Cc, U_ResNet_MiniImageNet as Cm
    sig = lambda c: list(inspect.signature(c.__init__).parameters)[1:]
    for c in (S.UResNet, S.UResNetNewEUCB, S.UResNetNoEUCB, S.UNet, S.UResNetNoResBlock):
        assert sig(c) == ["seg_classes", "in_channels", "base_c"]
        assert inspect.signature(c.__init__).parameters["base_c"].default == 32
    assert sig(S.EUCB2) == ["in_channels", "out_channels", "kernel_size", "stride", "activation", "scale_factor"]
    assert sig(S.BasicBlock) == ["inchannel", "outchannel", "stride"]
    assert sig(S.Up) == ["in_channels", "out_channels"]
    assert sig(Cc.UResNet4_No_SU) == ["num_classes", "in_channels", "ResidulBlock", "scale"]
    assert sig(Cm.U_ResNet) == ["num_classes", "in_channels", "ResidulBlock"]
    assert S.UResNetNewEUCB(19, 3, 8).name == "U-ResNetNewEUCB"


def test_mini_imagenet_state_dict_matches_oracle(oracle):
    from uresnet_amd.modules import U_ResNet_MiniImageNet as Cm
    a = Cm.U_ResNet(100, 3)
    b = oracle.MiniU_ResNet(100, 3)
    sa, sb = a.state_dict(), b.state_dict()
    assert list(sa.keys()) == list(sb.keys())
    assert all(sa[k].shape == sb[k].shape for k in sa)


def test_checkpoint_and_optimizer_state_interchange(tmp_path, oracle):
    """Checkpoints in the reference's dict format (train_UResNetAblitionStudy.py:244-252): a file written the way the
    reference writes it (torch.optim.Adam state, DataParallel `module.` prefix) loads here, and one written here loads
    into torch.optim.Adam and the reference-shaped model."""
    import uresnet_amd
    from uresnet_amd import train
    from uresnet_amd.modules.U_ResNet import UResNetNewEUCB
    torch.manual_seed(0)
    ref = oracle.UResNetNewEUCB(19, 3, 8)
    ropt = torch.optim.Adam(ref.parameters(), lr=1e-4)
    g = torch.Generator().manual_seed(3)
    x, t = torch.randn(1, 3, 32, 32, generator=g), torch.randint(0, 19, (1, 32, 32), generator=g)
    for _ in range(2):
        oracle.seg_train_step(ref, ropt, x, t)
    path = str(tmp_path / "ref_best_model.pth")
    torch.save({'epoch': 2, 'model_state_dict': {'module.' + k: v for k, v in ref.state_dict().items()},
                'optimizer_state_dict': ropt.state_dict(), 'val_loss': 1.5, 'metrics': {'iou': 0.25}}, path)
    m = UResNetNewEUCB(19, 3, 8)
    opt = train.Adam(m.parameters(), lr=1e-4)
    meta = train.load_checkpoint(path, m, opt)
    assert meta == {'epoch': 2, 'val_loss': 1.5, 'metrics': {'iou': 0.25}}
    for (k, a), (_, b) in zip(m.state_dict().items(), ref.state_dict().items()):
        assert torch.equal(a, b), k
    for p, rp in zip(m.parameters(), ref.parameters()):
        st, rst = opt.state[p], ropt.state[rp]
        assert int(st['step']) == 2 and st['step'].dtype == torch.int32 and st['step'].device == p.device   # counter beside the parameter
        assert torch.equal(st['exp_avg'], rst['exp_avg']) and torch.equal(st['exp_avg_sq'], rst['exp_avg_sq'])
    # and back: our file into the reference-shaped model and torch's optimiser
    path2 = str(tmp_path / "ours.pth")
    train.save_checkpoint(path2, m, opt, 3, 1.25, {'iou': 0.5})
    ck = torch.load(path2, weights_only=True)
    assert sorted(ck) == ['epoch', 'metrics', 'model_state_dict', 'optimizer_state_dict', 'val_loss']
    ref2 = oracle.UResNetNewEUCB(19, 3, 8)
    ref2.load_state_dict(ck['model_state_dict'], strict=True)
    ropt2 = torch.optim.Adam(ref2.parameters(), lr=1e-4)
    ropt2.load_state_dict(ck['optimizer_state_dict'])
    oracle.seg_train_step(ref2, ropt2, x, t)              # torch accepts the state and keeps stepping
    oracle.seg_train_step(ref, ropt, x, t)
    for a, b in zip(ref2.parameters(), ref.parameters()):
        assert torch.equal(a, b)
