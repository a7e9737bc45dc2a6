"""Time the BN backward-reduce kernel of several builds (tools/ablate/bn_*.so) on full-size activations."""
import ctypes as C
import glob
import os
import torch

SHAPES = [(6 * 512 * 1024, 64), (6 * 256 * 512, 128), (6 * 64 * 128, 512)]


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
                    C.c_longlong(rows), c, C.c_void_p(part.data_ptr()), st)
            runs.append((os.path.basename(path), lib, args, part))
        best = {}
        for rep in range(3):
            for name, lib, args, _ in runs:
                for _ in range(5):
                    assert lib.ur_bn_act_bwd_reduce(*args) == 0
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    lib.ur_bn_act_bwd_reduce(*args)
                e1.record()
                torch.cuda.synchronize()
                best[name] = min(best.get(name, 1e9), e0.elapsed_time(e1) / 20)
        for name, _, _, _ in runs:
            ms = best[name]
            print("rows %9d C %4d  %-14s %8.1f us  %6.2f TB/s" % (rows, c, name, ms * 1e3, 2.0 * rows * c * 2 / ms / 1e9), flush=True)


if __name__ == "__main__":
    main()
