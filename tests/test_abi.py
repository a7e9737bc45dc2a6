"""CPU-side checks of the drop-in boundary: the shared library loads, exports every symbol the
header declares, the ctypes table covers the header, and the model classes keep the
reference's constructor signatures and state_dict layout.  No kernel is launched here."""
import ctypes
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
