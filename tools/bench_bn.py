"""Time the BN backward-reduce kernel of several builds (tools/ablate/bn_*.so) on full-size activations."""
import ctypes as C
import glob
import os
import torch

SHAPES = [(6 * 512 * 1024, 32), (6 * 512 * 1024, 80), (6 * 512 * 1024, 64), (6 * 256 * 512, 128), (6 * 64 * 128, 512)]


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    libs = sorted(glob.glob(os.path.join(here, "ablate", "bn_*.so")))
    dev = torch.device("cuda:0")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for rows, c in SHAPES:
        dz = torch.randn(rows, c, device=dev).to(torch.bfloat16)
        y = torch.randn(rows, c, device=dev).to(torch.bfloat16)
        mean, invstd, gamma, beta = (torch.rand(c, device=dev) + 0.5 for _ in range(4))
        runs = []
        for path in libs:
            lib = C.CDLL(path)
            lib.ur_bn_stats_rows.restype = C.c_size_t
            lib.ur_bn_stats_rows.argtypes = [C.c_longlong, C.c_int]
            nrows = lib.ur_bn_stats_rows(rows, c)
            part = torch.empty(nrows, c, 2, device=dev)
            args = (C.c_void_p(dz.data_ptr()), C.c_void_p(y.data_ptr()), None, C.c_void_p(mean.data_ptr()),
                    C.c_void_p(invstd.data_ptr()), C.c_void_p(gamma.data_ptr()), C.c_void_p(beta.data_ptr()), 1, 1,
                    C.c_longlong(rows), c, 0, C.c_void_p(part.data_ptr()), st)
            z = torch.empty_like(y)
            fargs = (C.c_void_p(y.data_ptr()), C.c_void_p(mean.data_ptr()), C.c_void_p(invstd.data_ptr()),
                     C.c_void_p(gamma.data_ptr()), C.c_void_p(beta.data_ptr()), None, 1, C.c_void_p(z.data_ptr()), 1,
                     C.c_longlong(rows), c, 0, st)
            sums = torch.rand(c, 2, device=dev)
            aargs = (C.c_void_p(dz.data_ptr()), C.c_void_p(y.data_ptr()), None, C.c_void_p(mean.data_ptr()),
                     C.c_void_p(invstd.data_ptr()), C.c_void_p(gamma.data_ptr()), C.c_void_p(beta.data_ptr()),
                     C.c_void_p(sums.data_ptr()), 1, 0, C.c_void_p(z.data_ptr()), None, 1, C.c_longlong(rows), c, 0, st)
            sargs = (C.c_void_p(y.data_ptr()), 1, C.c_longlong(rows), c, C.c_void_p(part.data_ptr()), st)
            runs.append((os.path.basename(path), lib, args, part, fargs, aargs, z, sums, sargs))
        best = {}
        for rep in range(3):
            for r in runs:
                name, lib = r[0], r[1]
                for kind, fn, a, passes in (("reduce", lib.ur_bn_act_bwd_reduce, r[2], 2), ("fwd", lib.ur_bn_act_fwd, r[4], 2),
                                            ("apply", lib.ur_bn_act_bwd_apply, r[5], 3), ("stats", lib.ur_bn_stats, r[8], 1)):
                    for _ in range(5):
                        assert fn(*a) == 0
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(20):
                        fn(*a)
                    e1.record()
                    torch.cuda.synchronize()
                    key = (name, kind, passes)
                    best[key] = min(best.get(key, 1e9), e0.elapsed_time(e1) / 20)
        for (name, kind, passes), ms in best.items():
            print("rows %9d C %4d  %-14s %-7s %8.1f us  %6.2f TB/s" % (rows, c, name, kind, ms * 1e3, passes * rows * c * 2 / ms / 1e9), flush=True)


if __name__ == "__main__":
    main()
