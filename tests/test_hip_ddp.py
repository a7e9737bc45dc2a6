"""uresnet_amd.ddp.DataParallel over RCCL on the GPU: a one-rank process group exercises the real bucket / hook /
all-reduce path together with the fused HIP operators; three steps must track the plain single-GPU steps."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

SCRIPT = r'''
import os, sys, torch
import torch.distributed as dist
sys.path.insert(0, os.environ["UR_ROOT"])
import uresnet_amd
from uresnet_amd import train
from uresnet_amd.modules.U_ResNet import UResNetNewEUCB

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29541", rank=0, world_size=1, device_id=dev)
g = torch.Generator().manual_seed(7)
x = torch.randn(2, 3, 64, 96, generator=g).to(dev)
y = torch.randint(0, 19, (2, 64, 96), generator=g)
y[torch.rand(y.shape, generator=g) < 0.1] = 255
y = y.to(dev)

def run(use_ddp):
    torch.manual_seed(0)
    m = UResNetNewEUCB(19, 3, 8).set_compute_dtype(torch.float32).to(dev).train()
    opt = train.Adam(m.parameters(), lr=1e-3)
    ddp = uresnet_amd.ddp.DataParallel(m, bucket_mb=0.05) if use_ddp else None
    losses = []
    for _ in range(3):
        if ddp is None:
            loss, _ = train.seg_train_step(m, opt, x, y)
        else:
            loss, _ = train.seg_train_step(m, opt, x, y, count_hook=ddp.count_hook, grad_hook=ddp.finish)
        losses.append(loss.item())
    return losses, [p.detach().clone() for p in m.parameters()]

l0, p0 = run(False)
l1, p1 = run(True)
# the global-count path divides by an all-reduced count instead of averaging in the loss kernel: same value up to
# float rounding, which Adam then carries into the weights
assert all(abs(a - b) <= 1e-5 * abs(a) for a, b in zip(l0, l1)), (l0, l1)
for a, b in zip(p0, p1):
    assert (a - b).norm().item() <= 1e-3 * a.norm().item() + 1e-7
# the wrapper on one stream and on two (the gradients of a bucket then come from both streams and the collective waits
# for per-stream delivery events): same bits; every gradient arrives in its bucket slice without a copy
from uresnet_amd import ops
ops.branch_streams = False
l2, p2 = run(True)
ops.branch_streams = True
assert l1 == l2, (l1, l2)
for a, b in zip(p1, p2):
    assert torch.equal(a, b)
dist.destroy_process_group()
print("DDP_OK", l1)
'''


def test_rccl_one_rank_step_equals_plain_step():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, UR_ROOT=root, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0 and "DDP_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
