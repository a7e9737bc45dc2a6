"""The two kernels of the pair head's backward (ur_conv1x1_pw_wgrad_bn_pair, ur_conv1x1_pw_dgrad_bn_pair) on the default workload's
head (6 x 512 x 1024, 64 + 16 channels -> 19 classes, gradient rows widened to 24) and on the EUCB* projection (32 -> 16):
    python tools/bench_head_bwd.py [a.so b.so ...]      (default: the in-tree library)
Libraries are interleaved; best of three rounds of 20 launches."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import uresnet_amd  # noqa: E402,F401
from uresnet_amd import _lib as L  # noqa: E402

dev = "cuda:0"
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731


def timed(f):
    for _ in range(3):
        f()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
    return best


def main():
    paths = sys.argv[1:] or [L.lib_path() if hasattr(L, "lib_path") else os.path.join(os.path.dirname(L.__file__), "lib", "liburesnet_hip.so")]
    libs = [(os.path.basename(p), C.CDLL(os.path.abspath(p))) for p in paths]
    for (n, h, w, ca, cb, cout, cpad) in [(6, 512, 1024, 64, 16, 19, 24), (6, 512, 1024, 32, 0, 16, 16)]:
        c = ca + cb
        ya = torch.randn(n, h, w, ca, device=dev).to(torch.bfloat16)
        yb = torch.randn(n, h, w, cb, device=dev).to(torch.bfloat16) if cb else None
        dz = torch.randn(n, h, w, cpad, device=dev).to(torch.bfloat16)
        wt = torch.randn(cout, c, device=dev) * 0.1
        mean, invstd = torch.randn(c, device=dev) * 0.1, torch.rand(c, device=dev) + 0.5
        gamma, beta = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.3
        k = gamma * invstd
        affs = [torch.cat([k[:ca], (beta - mean * k)[:ca]]).contiguous(), torch.cat([k[ca:], (beta - mean * k)[ca:]]).contiguous()]
        d = L.ConvDesc(L.UR_BF16, n, h, w, c, h, w, cpad, 1, 1, 1, 0, c, cpad)
        dw, sums = torch.empty(cout, c, device=dev), torch.empty(c, 2, device=dev)
        dga, dbe = torch.empty(c, device=dev), torch.empty(c, device=dev)
        dya = torch.empty_like(ya)
        dyb = torch.empty_like(yb) if cb else None
        for name, lib in libs:
            lib.ur_conv1x1_pw_wgrad_bn_workspace.restype = C.c_size_t
            ws = torch.empty(lib.ur_conv1x1_pw_wgrad_bn_workspace(C.byref(d)), dtype=torch.uint8, device=dev)
            if cb:
                wg = lambda: lib.ur_conv1x1_pw_wgrad_bn_pair(vp(ya), ca, ca, vp(yb), cb, cb, vp(affs[0]), vp(affs[1]), vp(mean), vp(invstd),  # noqa: E731
                                                             vp(gamma), vp(beta), vp(dz), vp(wt), cout, vp(dw), vp(sums), vp(dga), vp(dbe),
                                                             vp(ws), C.c_size_t(ws.numel()), C.byref(d), st)
            else:
                wg = lambda: lib.ur_conv1x1_pw_wgrad_bn(vp(ya), vp(affs[0]), vp(mean), vp(invstd), vp(gamma), vp(beta), vp(dz), vp(wt), cout,  # noqa: E731
                                                        vp(dw), vp(sums), vp(dga), vp(dbe), vp(ws), C.c_size_t(ws.numel()), C.byref(d), st)
            dg = lambda: lib.ur_conv1x1_pw_dgrad_bn_pair(vp(dz), vp(wt), cout, c, vp(ya), vp(dya), ca, ca, vp(yb), vp(dyb), cb, cb, vp(mean),  # noqa: E731
                                                         vp(invstd), vp(gamma), vp(beta), vp(sums), C.byref(d), st)
            assert wg() == 0 and dg() == 0
            tw, td = timed(wg), timed(dg)
            px = n * h * w
            bw, bd = px * (c + cpad) * 2, px * (cpad + 2 * c) * 2
            print("%-22s %s  wgrad+sums %.1f us (%.2f TB/s)   dgrad+apply %.1f us (%.2f TB/s)  check %.6e %.6e" % (
                name, (n, h, w, ca, cb, cout), tw, bw / tw / 1e6, td, bd / td / 1e6, sums.double().abs().sum().item(),
                dya.double().abs().sum().item()), flush=True)


main()
