"""Training throughput of the other configurations and ablation variants (bench.py covers the headline one).

    python tools/bench_configs.py            # one line per case: ms per step, images/s

Cases: the CIFAR-shaped and 224x224 classification nets (SGD), the Cityscapes ablation variants at 6x3x512x1024
(Adam 1e-4, CE ignore 255), the headline net at 1024x2048, and the scratch variants of temp.py.  Synthetic inputs
(seed 1234), default seeded initialisation, bf16 activations unless stated, inputs resident in HBM.
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import uresnet_amd  # noqa: E402
from uresnet_amd import train  # noqa: E402
from uresnet_amd.modules import U_ResNet as S, U_ResNet_CIFAR as CC, U_ResNet_MiniImageNet as CM, temp as CT  # noqa: E402

DEV = torch.device("cuda:0")


def timed(step, warmup=3, steps=10):
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def seg_case(name, build, shape, classes=19, dtype=torch.bfloat16):
    torch.manual_seed(0)
    model = build().set_compute_dtype(dtype).to(DEV).train()
    opt = train.Adam(model.parameters(), lr=1e-4)
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(*shape, generator=g).to(DEV)
    if classes > 1:
        y = torch.randint(0, classes, (shape[0], shape[2], shape[3]), generator=g)
        y[torch.rand(y.shape, generator=g) < 0.1] = 255
        y = y.to(DEV)
        dt = timed(lambda: train.seg_train_step(model, opt, x, y))
    else:
        y = (torch.rand(shape[0], 1, shape[2], shape[3], generator=g) > 0.5).float().to(DEV)
        dt = timed(lambda: train.ice_train_step(model, opt, x, y))
    report(name, shape, dtype, dt)


def cls_case(name, build, shape, classes, lr, dtype=torch.bfloat16, graph=False):
    torch.manual_seed(0)
    model = build().set_compute_dtype(dtype).to(DEV).train()
    opt = train.SGD(model.parameters(), lr=lr)
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(*shape, generator=g).to(DEV)
    y = torch.randint(0, classes, (shape[0],), generator=g).to(DEV)
    if graph:                                   # the whole step as one HIP graph replay (launch-bound shapes)
        step = train.GraphedStep(train.cls_train_step, model, opt, x, y, warmup=3)
        dt = timed(lambda: step(x, y))
        name += " graph"
    else:
        dt = timed(lambda: train.cls_train_step(model, opt, x, y))
    report(name, shape, dtype, dt)


def report(name, shape, dtype, dt):
    print("%-34s %-18s %-5s %8.2f ms/step %9.1f images/s" % (name, "x".join(map(str, shape)),
                                                             "bf16" if dtype == torch.bfloat16 else "fp32", dt * 1e3, shape[0] / dt),
          flush=True)
    torch.cuda.empty_cache()


def main():
    uresnet_amd._lib.load()
    cls_case("UResNet4_No_SU(10,3,scale=3)", lambda: CC.UResNet4_No_SU(10, 3, scale=3), (64, 3, 32, 32), 10, 0.01, torch.float32)
    cls_case("UResNet4_No_SU(10,3,scale=3)", lambda: CC.UResNet4_No_SU(10, 3, scale=3), (64, 3, 32, 32), 10, 0.01)
    cls_case("UResNet4_No_SU(10,3,scale=3)", lambda: CC.UResNet4_No_SU(10, 3, scale=3), (64, 3, 32, 32), 10, 0.01, graph=True)
    cls_case("UResNet4_No_SU(10,3,scale=3)", lambda: CC.UResNet4_No_SU(10, 3, scale=3), (256, 3, 32, 32), 10, 0.01, graph=True)
    cls_case("mini U_ResNet(100,3)", lambda: CM.U_ResNet(100, 3), (32, 3, 224, 224), 100, 0.1)
    cls_case("mini U_ResNet(100,3)", lambda: CM.U_ResNet(100, 3), (32, 3, 224, 224), 100, 0.1, graph=True)
    cls_case("temp NlNet(100,3)", lambda: CT.NlNet(100, 3), (32, 3, 224, 224), 100, 0.01)
    cls_case("temp UResNet2(100,3)", lambda: CT.UResNet2(100, 3), (32, 3, 224, 224), 100, 0.01)
    seg_case("UResNetNewEUCB(19,3,64)", lambda: S.UResNetNewEUCB(19, 3, 64), (6, 3, 512, 1024))
    seg_case("UResNet(19,3,64)", lambda: S.UResNet(19, 3, 64), (6, 3, 512, 1024))
    seg_case("UResNetNoEUCB(19,3,64)", lambda: S.UResNetNoEUCB(19, 3, 64), (6, 3, 512, 1024))
    seg_case("UNet(19,3,64)", lambda: S.UNet(19, 3, 64), (6, 3, 512, 1024))
    seg_case("UResNetNoResBlock(1,3,64) BCE", lambda: S.UResNetNoResBlock(1, 3, 64), (6, 3, 512, 1024), classes=1)
    seg_case("UResNetNewEUCB(19,3,64)", lambda: S.UResNetNewEUCB(19, 3, 64), (2, 3, 1024, 2048))
    seg_case("UResNetNewEUCB(19,3,64)", lambda: S.UResNetNewEUCB(19, 3, 64), (16, 3, 512, 1024))


if __name__ == "__main__":
    main()
