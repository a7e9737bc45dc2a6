"""Time per launch of the 2x2 stride-2 transposed convolutions of the network (forward and backward-data, the implicit-GEMM
kernel) through the C ABI.  python tools/bench_convt.py [lib.so]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import uresnet_amd  # noqa: E402,F401
from uresnet_amd._lib import ConvDesc, UR_BF16, PACK_CONVT_FWD, PACK_CONVT_DGRAD  # noqa: E402

SHAPES = [(6, 256, 512, 128, 64), (6, 128, 256, 256, 128), (6, 64, 128, 512, 256), (6, 32, 64, 1024, 512)]


def timed(fn, reps=20):
    best = 1e9
    for _ in range(3):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "u-resnet_amd", "lib", "liburesnet_hip.so")
    lib = C.CDLL(os.path.abspath(path))
    lib.ur_conv2d_packed_elems.restype = C.c_size_t
    dev = torch.device("cuda:0")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    vp = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    for (n, h, w, cin, cout) in SHAPES:
        torch.manual_seed(0)
        x = torch.randn(n, h, w, cin, device=dev).to(torch.bfloat16)
        dy = torch.randn(n, 2 * h, 2 * w, cout, device=dev).to(torch.bfloat16)
        wt = torch.randn(cin, cout, 2, 2, device=dev) * (1.0 / cin) ** 0.5
        bias = torch.randn(cout, device=dev)
        d = ConvDesc(UR_BF16, n, h, w, cin, 2 * h, 2 * w, cout, 2, 2, 2, 0, cin, cout)
        flops = 2.0 * n * h * w * cin * cout * 4
        wf = torch.empty(lib.ur_conv2d_packed_elems(C.byref(d), PACK_CONVT_FWD), dtype=torch.bfloat16, device=dev)
        wd = torch.empty(lib.ur_conv2d_packed_elems(C.byref(d), PACK_CONVT_DGRAD), dtype=torch.bfloat16, device=dev)
        assert lib.ur_conv2d_pack_weight(vp(wt), vp(wf), C.byref(d), PACK_CONVT_FWD, st) == 0
        assert lib.ur_conv2d_pack_weight(vp(wt), vp(wd), C.byref(d), PACK_CONVT_DGRAD, st) == 0
        y = torch.empty(n, 2 * h, 2 * w, cout, dtype=torch.bfloat16, device=dev)
        dx = torch.empty(n, h, w, cin, dtype=torch.bfloat16, device=dev)
        f = lambda: lib.ur_convT2x2_fwd(vp(x), vp(wf), vp(bias), vp(y), C.byref(d), st)  # noqa: E731
        g = lambda: lib.ur_convT2x2_dgrad(vp(dy), vp(wd), vp(dx), C.byref(d), st)  # noqa: E731
        lib.ur_convT2x2_wgrad_workspace.restype = C.c_size_t
        ws = torch.empty(lib.ur_convT2x2_wgrad_workspace(C.byref(d)), dtype=torch.uint8, device=dev)
        dw = torch.empty(cin, cout, 2, 2, device=dev)
        wg = lambda: lib.ur_convT2x2_wgrad(vp(x), vp(dy), vp(dw), vp(ws), C.c_size_t(ws.numel()), C.byref(d), st)  # noqa: E731
        assert f() == 0 and g() == 0 and wg() == 0
        torch.cuda.synchronize()
        ref = torch.nn.functional.conv_transpose2d(x.float().permute(0, 3, 1, 2), wt.to(torch.bfloat16).float(), bias, stride=2)
        err = ((y.float().permute(0, 3, 1, 2) - ref).norm() / ref.norm()).item()
        refw = torch.einsum("nyxi,nyxqo->ioq", x.float(), dy.float().view(n, h, 2, w, 2, cout).permute(0, 1, 3, 2, 4, 5).reshape(
            n, h, w, 4, cout)).reshape(cin, cout, 2, 2)
        errw = ((dw - refw).norm() / refw.norm()).item()
        tf, tg, tw = timed(f), timed(g), timed(wg)
        print("%-26s fwd %7.1f us %6.0f TF (err %.1e)   dgrad %7.1f us %6.0f TF   wgrad %7.1f us %6.0f TF (err %.1e)" % (
            (n, h, w, cin, cout), tf * 1e3, flops / tf / 1e9, err, tg * 1e3, flops / tg / 1e9, tw * 1e3, flops / tw / 1e9, errw),
            flush=True)


if __name__ == "__main__":
    main()
