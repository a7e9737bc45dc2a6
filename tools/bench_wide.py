"""The persistent one-workgroup-per-CU 3x3 kernel (conv3x3_wide.hip) against the two-workgroups-per-CU tile kernel:
results on the same operands (both against an fp32 convolution of the bf16-rounded operands) and time per launch.

python tools/bench_wide.py [lib.so]; WD_SHAPE=<index> restricts the run to one shape; WD_ZERO=1 uses all-zero operands.
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import uresnet_amd  # noqa: E402,F401
from uresnet_amd._lib import ConvDesc, UR_BF16  # noqa: E402

SHAPES = [(6, 256, 512, 128, 128), (6, 128, 256, 256, 256), (6, 64, 128, 512, 512), (6, 256, 512, 64, 128), (6, 256, 512, 256, 128),
          (6, 128, 256, 128, 256), (6, 32, 64, 1024, 1024), (2, 37, 50, 64, 128), (1, 16, 32, 32, 128)]


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
    shapes = SHAPES
    if os.environ.get("WD_SHAPE"):
        shapes = [SHAPES[int(os.environ["WD_SHAPE"])]]
    dev = torch.device("cuda:0")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    lib = C.CDLL(os.path.abspath(path))
    for f in ("ur_conv3x3_tile_packed_elems", "ur_conv3x3_wide_packed_elems"):
        getattr(lib, f).restype = C.c_size_t
    vp = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    for (n, h, w, cin, cout) in shapes:
        torch.manual_seed(0)
        x = torch.randn(n, h, w, cin, device=dev).relu_().to(torch.bfloat16)
        wt = (torch.randn(cout, cin, 3, 3, device=dev) * (2.0 / (9 * cin)) ** 0.5)
        if os.environ.get("WD_ZERO"):
            x.zero_()
        d = ConvDesc(UR_BF16, n, h, w, cin, h, w, cout, 3, 3, 1, 1, cin, cout)
        flops = 2.0 * n * h * w * cin * cout * 9
        line = "%-26s" % ((n, h, w, cin, cout),)
        for dgrad in (0, 1):
            src = x if not dgrad else torch.randn(n, h, w, cout, device=dev).to(torch.bfloat16)
            oc = cout if not dgrad else cin
            if dgrad:
                ref = torch.nn.grad.conv2d_input((n, cin, h, w), wt.to(torch.bfloat16).float(), src.float().permute(0, 3, 1, 2),
                                                 padding=1) if n * h * w * cin <= 2 ** 28 else None
            else:
                ref = torch.nn.functional.conv2d(src.float().permute(0, 3, 1, 2), wt.to(torch.bfloat16).float(),
                                                 padding=1) if n * h * w * cout <= 2 ** 28 else None
            res = {}
            for name in ("tile", "wide"):
                if not getattr(lib, "ur_conv3x3_%s_supported" % name)(C.byref(d), dgrad):
                    res[name] = None
                    continue
                wpk = torch.empty(getattr(lib, "ur_conv3x3_%s_packed_elems" % name)(C.byref(d)), dtype=torch.bfloat16, device=dev)
                assert getattr(lib, "ur_conv3x3_%s_pack_weight" % name)(vp(wt), vp(wpk), C.byref(d), dgrad, st) == 0
                y = torch.full((n, h, w, oc), float("nan"), dtype=torch.bfloat16, device=dev)
                rows = getattr(lib, "ur_conv3x3_%s_stats_rows" % name)(C.byref(d))
                stats = torch.zeros(rows, cout, 2, device=dev)
                if dgrad:
                    call = lambda: getattr(lib, "ur_conv3x3_%s_dgrad" % name)(vp(src), vp(wpk), vp(y), C.byref(d), st)  # noqa: E731
                else:
                    call = lambda: getattr(lib, "ur_conv3x3_%s_fwd" % name)(vp(src), vp(wpk), vp(y), vp(stats), C.byref(d), st)  # noqa: E731
                assert call() == 0
                torch.cuda.synchronize()
                res[name] = (y, stats.sum(0), timed(call))
            for name in ("tile", "wide"):
                if res[name] is None:
                    line += "  %s %s: unsupported" % ("dgrad" if dgrad else "fwd", name)
                    continue
                y, ssum, t = res[name]
                err = float("nan")
                if ref is not None:
                    r = ref.permute(0, 2, 3, 1)
                    err = ((y.float() - r).norm() / r.norm().clamp_min(1e-30)).item()
                line += "  %s %s %7.1f us %6.0f TF err %.1e" % ("dgrad" if dgrad else "fwd", name, t * 1e3, flops / t / 1e9, err)
                if not dgrad and ref is not None:
                    r = ref.permute(0, 2, 3, 1).reshape(-1, cout)
                    s_ref = torch.stack([r.sum(0), (r * r).sum(0)], 1)
                    line += " stats %.1e" % ((ssum - s_ref).norm() / s_ref.norm()).item()
            if res["tile"] is not None and res["wide"] is not None:
                line += "  | max|wide-tile| %.2e" % (res["wide"][0].float() - res["tile"][0].float()).abs().max().item()
        print(line, flush=True)
        # backward-data epilogue variants (addend; BatchNorm-backward sums with the mask rebuilt from the BatchNorm's input)
        if lib.ur_conv3x3_wide_supported(C.byref(d), 1):
            dyt = torch.randn(n, h, w, cout, device=dev).to(torch.bfloat16)
            add = torch.randn(n, h, w, cin, device=dev).to(torch.bfloat16)
            bny = torch.randn(n, h, w, cin, device=dev).to(torch.bfloat16)
            mean, inv = torch.randn(cin, device=dev) * 0.1, torch.rand(cin, device=dev) + 0.5
            gam, bet = torch.randn(cin, device=dev), torch.randn(cin, device=dev) * 0.3
            out, tms = {}, {}
            for name in ("tile", "wide"):
                wpk = torch.empty(getattr(lib, "ur_conv3x3_%s_packed_elems" % name)(C.byref(d)), dtype=torch.bfloat16, device=dev)
                assert getattr(lib, "ur_conv3x3_%s_pack_weight" % name)(vp(wt), vp(wpk), C.byref(d), 1, st) == 0
                dx1 = torch.full((n, h, w, cin), float("nan"), dtype=torch.bfloat16, device=dev)
                dx2 = torch.full((n, h, w, cin), float("nan"), dtype=torch.bfloat16, device=dev)
                rows = (lib.ur_conv3x3_tile_dgrad_stats_rows if name == "tile" else lib.ur_conv3x3_wide_stats_rows)(C.byref(d))
                part = torch.zeros(rows, cin, 2, device=dev)
                f_add = lambda: getattr(lib, "ur_conv3x3_%s_dgrad_add" % name)(vp(dyt), vp(wpk), vp(dx1), vp(add), cin, C.byref(d), st)  # noqa: E731
                if name == "tile":
                    f_bn = lambda: lib.ur_conv3x3_tile_dgrad_bnsums(vp(dyt), vp(wpk), vp(dx2), None, 0, vp(bny), cin, None, vp(mean),  # noqa: E731
                                                                    vp(inv), vp(gam), vp(bet), vp(part), C.byref(d), st)
                else:
                    f_bn = lambda: lib.ur_conv3x3_wide_dgrad_bnsums(vp(dyt), vp(wpk), vp(dx2), vp(bny), cin, vp(mean), vp(inv), vp(gam),  # noqa: E731
                                                                    vp(bet), vp(part), C.byref(d), st)
                assert f_add() == 0 and f_bn() == 0
                torch.cuda.synchronize()
                out[name] = (dx1, dx2, part.sum(0))
                tms[name] = (timed(f_add), timed(f_bn))
            t_, w_ = out["tile"], out["wide"]
            print("    dgrad+add  tile %7.1f us wide %7.1f us  max|diff| %.2e   dgrad+bnsums tile %7.1f us wide %7.1f us  max|dx diff| %.2e  sums rel %.1e" % (
                tms["tile"][0] * 1e3, tms["wide"][0] * 1e3, (t_[0].float() - w_[0].float()).abs().max().item(),
                tms["tile"][1] * 1e3, tms["wide"][1] * 1e3, (t_[1].float() - w_[1].float()).abs().max().item(),
                ((t_[2] - w_[2]).norm() / t_[2].norm()).item()), flush=True)


if __name__ == "__main__":
    main()
