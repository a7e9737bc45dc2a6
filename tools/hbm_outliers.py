"""Every launch of the memory-bound passes of one training step of the default workload, one stream, with its algorithmic
bytes and event-measured time: the launches furthest below the streaming rate come first.

    python tools/hbm_outliers.py [rate TB/s the excess is priced against, default 5.8]
"""
import os
import sys

import torch

os.environ.setdefault("URESNET_BRANCH_STREAMS", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import uresnet_amd as pkg  # noqa: E402


def main():
    rate = float(sys.argv[1]) if len(sys.argv) > 1 else 5.8
    M, train, ops = pkg.modules.U_ResNet, pkg.train, pkg.ops
    dev = "cuda:0"
    torch.manual_seed(0)
    m = M.UResNetNewEUCB(19, 3, 64).set_compute_dtype(torch.bfloat16).to(dev).train()
    opt = train.Adam(m.parameters(), lr=1e-4)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(6, 3, 512, 1024, generator=g).to(dev)
    y = torch.randint(0, 19, (6, 512, 1024), generator=g).to(dev)
    for _ in range(3):
        train.seg_train_step(m, opt, x, y)
    torch.cuda.synchronize()
    ops.kernel_timings, ops.kernel_timing_tags = [], None
    train.seg_train_step(m, opt, x, y)
    torch.cuda.synchronize()
    tm, ops.kernel_timings = ops.kernel_timings, None
    rows = []
    for i, (tag, work, e0, e1, shape) in enumerate(tm):
        if not tag.startswith("hbm:"):
            continue
        ms = e0.elapsed_time(e1)
        rows.append((ms - work / (rate * 1e9), i, tag, work, ms))
    print("%d memory-bound launches, %.2f ms in total, %.2f ms above %.1f TB/s" % (
        len(rows), sum(r[4] for r in rows), sum(max(r[0], 0.0) for r in rows), rate))
    for excess, i, tag, work, ms in sorted(rows, reverse=True)[:40]:
        print("#%-4d %-22s %8.1f MB %8.1f us  %5.2f TB/s  excess %6.1f us" % (i, tag, work / 1e6, ms * 1e3, work / ms / 1e9, excess * 1e3))


if __name__ == "__main__":
    main()
