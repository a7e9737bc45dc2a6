"""Inference throughput (model.eval() under no_grad): BatchNorm folded into the convolution epilogue vs separate passes.

python tools/bench_eval.py            # UResNetNewEUCB(19,3,64) at 6x3x512x1024 and UResNet(1,3,64) at 32x3x512x512, bf16
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import uresnet_amd  # noqa: E402
from uresnet_amd import nn as hnn  # noqa: E402
from uresnet_amd.modules.U_ResNet import UResNet, UResNetNewEUCB  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    cases = [("UResNetNewEUCB(19,3,64) 6x3x512x1024", UResNetNewEUCB(19, 3, 64), (6, 3, 512, 1024)),
             ("UResNet(1,3,64) 32x3x512x512", UResNet(1, 3, 64), (32, 3, 512, 512))]
    for name, model, shape in cases:
        model = model.set_compute_dtype(torch.bfloat16).to(dev).eval()
        x = torch.randn(*shape, device=dev)
        best = {}
        with torch.no_grad():
            for rep in range(3):
                for fused in (False, True):
                    hnn.fuse_eval_bn = fused
                    for _ in range(3):
                        model(x)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(10):
                        y = model(x)
                    torch.cuda.synchronize()
                    dt = (time.perf_counter() - t0) / 10
                    best[fused] = min(best.get(fused, 1e9), dt)
        hnn.fuse_eval_bn = True
        for fused in (False, True):
            print("%-40s BN %-8s %7.2f ms  %8.1f images/s" % (name, "folded" if fused else "separate", best[fused] * 1e3,
                                                              shape[0] / best[fused]), flush=True)
        del model, x, y
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
