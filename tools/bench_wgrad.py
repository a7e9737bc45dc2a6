"""Timing of the 3x3 backward-weight tile kernel (version from URESNET_WGRAD_V) on the network's layer shapes.

python tools/bench_wgrad.py [lib.so]; WG_SHAPE=<index> restricts the run to one shape (hardware-counter passes).
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import uresnet_amd  # noqa: E402,F401
from uresnet_amd._lib import ConvDesc, UR_BF16  # noqa: E402

SHAPES = [(6, 256, 512, 128, 128), (6, 512, 1024, 64, 64), (6, 64, 128, 512, 512), (6, 128, 256, 256, 256),
          (6, 256, 512, 64, 64), (6, 32, 64, 1024, 1024), (2, 37, 50, 64, 128)]


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "u-resnet_amd", "lib", "liburesnet_hip.so")
    if os.environ.get("WG_SHAPE"):
        SHAPES[:] = [SHAPES[int(os.environ["WG_SHAPE"])]]
    dev = torch.device("cuda:0")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    lib = C.CDLL(os.path.abspath(path))
    lib.ur_wgrad3x3_tile_workspace.restype = C.c_size_t
    for (n, h, w, cin, cout) in SHAPES:
        x = torch.randn(n, h, w, cin, device=dev).relu_().to(torch.bfloat16)
        dy = torch.randn(n, h, w, cout, device=dev).to(torch.bfloat16)
        if os.environ.get("WG_ZERO"):            # all-zero operands: the same instruction stream at a fraction of the power
            x.zero_(); dy.zero_()
        dw = torch.empty(cout, cin, 3, 3, device=dev)
        d = ConvDesc(UR_BF16, n, h, w, cin, h, w, cout, 3, 3, 1, 1, cin, cout)
        nws = lib.ur_wgrad3x3_tile_workspace(C.byref(d))
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        args = (C.c_void_p(x.data_ptr()), C.c_void_p(dy.data_ptr()), C.c_void_p(dw.data_ptr()), C.c_void_p(ws.data_ptr()),
                C.c_size_t(nws), C.byref(d), st)
        flops = 2.0 * n * h * w * cin * cout * 9
        # correctness against the generic backward-weight kernel (an independent implementation) on the same operands
        lib.ur_conv2d_wgrad_workspace.restype = C.c_size_t
        nws2 = lib.ur_conv2d_wgrad_workspace(C.byref(d))
        ws2 = torch.empty(max(nws2, 16), dtype=torch.uint8, device=dev)
        ref = torch.empty_like(dw)
        assert lib.ur_conv2d_wgrad(C.c_void_p(x.data_ptr()), C.c_void_p(dy.data_ptr()), C.c_void_p(ref.data_ptr()),
                                   C.c_void_p(ws2.data_ptr()), C.c_size_t(nws2), C.byref(d), st) == 0
        assert lib.ur_wgrad3x3_tile(*args) == 0
        torch.cuda.synchronize()
        err = ((dw - ref).double().norm() / ref.double().norm().clamp_min(1e-30)).item()
        best = 1e9
        for rep in range(3):
            for _ in range(10):
                assert lib.ur_wgrad3x3_tile(*args) == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                lib.ur_wgrad3x3_tile(*args)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20)
        print("%-28s v%s %8.1f us  %7.1f TF (tile kernel + slab reduce)  rel-L2 vs generic kernel %.1e" % (
            (n, h, w, cin, cout), os.environ.get("URESNET_WGRAD_V", "1"), best * 1e3, flops / best / 1e9, err), flush=True)


if __name__ == "__main__":
    main()
