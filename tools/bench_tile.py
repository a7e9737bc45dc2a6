"""A/B timing of builds of the 3x3 tile kernel: python tools/bench_tile.py a.so b.so ...

Each library must export the ur_conv3x3_tile_* entry points.  Variants are interleaved and the best of three
rounds (40 launches each after 20 warm-up launches) is kept, so that clock ramps do not bias the comparison.
TILE_MODE=dgrad_bn times ur_conv3x3_tile_dgrad_bnsums (backward-data + BatchNorm-backward sums) instead of the forward.
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import uresnet_amd  # noqa: E402,F401
from uresnet_amd._lib import ConvDesc, UR_BF16  # noqa: E402

SHAPES = [(6, 256, 512, 128, 128), (6, 256, 512, 64, 64), (6, 64, 128, 512, 512), (6, 512, 1024, 64, 64),
          (6, 128, 256, 256, 256)]


def main():
    libs = sys.argv[1:]
    if os.environ.get("TILE_SHAPE"):          # one shape only (hardware-counter runs)
        SHAPES[:] = [SHAPES[int(os.environ["TILE_SHAPE"])]]
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    for (n, h, w, cin, cout) in SHAPES:
        x = torch.randn(n, h, w, cin, device=dev).relu_().to(torch.bfloat16)
        wt = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
        if os.environ.get("TILE_ZERO"):           # all-zero operands: the same instruction stream at a fraction of the power
            x.zero_(); wt.zero_()
        y = torch.empty(n, h, w, cout, device=dev, dtype=torch.bfloat16)
        d = ConvDesc(UR_BF16, n, h, w, cin, h, w, cout, 3, 3, 1, 1, cin, cout)
        flops = 2.0 * n * h * w * cin * cout * 9
        runs, ref, keep = [], None, []
        for path in libs:
            lib = C.CDLL(os.path.abspath(path))
            lib.ur_conv3x3_tile_packed_elems.restype = C.c_size_t
            wp = torch.empty(lib.ur_conv3x3_tile_packed_elems(C.byref(d)), device=dev, dtype=torch.bfloat16)
            lib.ur_conv3x3_tile_pack_weight(C.c_void_p(wt.data_ptr()), C.c_void_p(wp.data_ptr()), C.byref(d), 0, C.c_void_p(st))
            if os.environ.get("TILE_MODE") == "dgrad_bn":
                if not keep:                  # the same operands for every library: the kernel reduces over cout, produces cin
                    keep.append((torch.randn(n, h, w, cout, device=dev).to(torch.bfloat16),
                                 torch.randn(n, h, w, cin, device=dev).to(torch.bfloat16),
                                 [torch.rand(cin, device=dev) + 0.5 for _ in range(4)]))
                dyt, bny, vec = keep[0]
                dxt = torch.empty(n, h, w, cin, device=dev, dtype=torch.bfloat16)
                lib.ur_conv3x3_tile_pack_weight(C.c_void_p(wt.data_ptr()), C.c_void_p(wp.data_ptr()), C.byref(d), 1, C.c_void_p(st))
                rows = lib.ur_conv3x3_tile_dgrad_stats_rows(C.byref(d))
                part = torch.zeros(rows, cin, 2, device=dev)
                keep.append((dxt, part))
                args = (C.c_void_p(dyt.data_ptr()), C.c_void_p(wp.data_ptr()), C.c_void_p(dxt.data_ptr()), None, 0,
                        C.c_void_p(bny.data_ptr()), 0, None, C.c_void_p(vec[0].data_ptr()), C.c_void_p(vec[1].data_ptr()),
                        C.c_void_p(vec[2].data_ptr()), C.c_void_p(vec[3].data_ptr()), C.c_void_p(part.data_ptr()), C.byref(d),
                        C.c_void_p(st))
                fn = lib.ur_conv3x3_tile_dgrad_bnsums
                assert fn(*args) == 0
                out = torch.cat([dxt.float().flatten(), part.flatten() * 1e-3]).clone()
                if ref is None:
                    ref = out
                err = float((out - ref).norm() / ref.norm())
                runs.append((os.path.basename(path), fn, args, wp, err))
                continue
            args = (C.c_void_p(x.data_ptr()), C.c_void_p(wp.data_ptr()), C.c_void_p(y.data_ptr()), None, C.byref(d), C.c_void_p(st))
            assert lib.ur_conv3x3_tile_fwd(*args) == 0
            out = y.float().clone()
            if ref is None:
                ref = out
            err = float((out - ref).norm() / ref.norm())
            runs.append((os.path.basename(path), lib.ur_conv3x3_tile_fwd, args, wp, err))
        best = {}
        for rep in range(3):
            for name, fn, args, _, _ in runs:
                for _ in range(20):
                    fn(*args)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(40):
                    fn(*args)
                e1.record()
                torch.cuda.synchronize()
                best[name] = min(best.get(name, 1e9), e0.elapsed_time(e1) / 40)
        for name, _, _, _, err in runs:
            ms = best[name]
            print("%-28s %-14s %8.1f us  %7.1f TF  rel.diff vs first %.1e" % ((n, h, w, cin, cout), name, ms * 1e3, flops / ms / 1e9, err), flush=True)


if __name__ == "__main__":
    main()
