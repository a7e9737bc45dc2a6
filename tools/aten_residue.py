"""Which ATen kernels (fills, copies) still run inside a training step of the default workload, and from where.

python tools/aten_residue.py: runs three steps, profiles the fourth with torch.profiler and prints every aten op that
launched a device kernel, with its input shapes and the python frames above it."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import uresnet_amd  # noqa: E402
from uresnet_amd import train  # noqa: E402
from uresnet_amd.modules.U_ResNet import UResNetNewEUCB  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    uresnet_amd._lib.load()
    torch.manual_seed(0)
    model = UResNetNewEUCB(19, 3, 64).set_compute_dtype(torch.bfloat16).to(dev).train()
    opt = train.Adam(model.parameters(), lr=1e-4)
    b, h, w = 2, 256, 512
    x = torch.randn(b, 3, h, w, device=dev)
    y = torch.randint(0, 19, (b, h, w), device=dev)
    for _ in range(3):
        train.seg_train_step(model, opt, x, y)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        train.seg_train_step(model, opt, x, y)
        torch.cuda.synchronize()
    seen = {}
    for ev in prof.events():
        if not ev.name.startswith("aten::") or ev.device_time_total <= 0:
            continue
        if ev.cpu_children and any(c.name.startswith("aten::") and c.device_time_total > 0 for c in ev.cpu_children):
            continue                                   # report the innermost aten op that owns the kernel
        frames = [f for f in (ev.stack or []) if "uresnet_amd" in f or "u-resnet_amd" in f or "autograd" in f][:3]
        key = (ev.name, str(ev.input_shapes), tuple(frames))
        seen[key] = seen.get(key, 0) + 1
    for (name, shapes, frames), n in sorted(seen.items(), key=lambda kv: -kv[1]):
        print("%3d x %-18s %s" % (n, name, shapes[:100]))
        for f in frames:
            print("        ", f[:150])


if __name__ == "__main__":
    main()
