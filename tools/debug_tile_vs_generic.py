import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import uresnet_amd
from uresnet_amd import ops
H, W = int(sys.argv[1]) if len(sys.argv) > 1 else 512, int(sys.argv[2]) if len(sys.argv) > 2 else 1024
g = torch.Generator().manual_seed(1234)
x = torch.randn(1, 3, H, W, generator=g).cuda()
y = torch.randint(0, 19, (1, H, W), generator=g)
y[torch.rand(y.shape, generator=g) < 0.1] = 255
y = y.cuda()
res = {}
for tile in (True, False):
    ops.use_tile_kernel = tile
    torch.manual_seed(0)
    m = uresnet_amd.modules.U_ResNet.UResNetNewEUCB(19, 3, 64).set_compute_dtype(torch.bfloat16).cuda().train()
    out = m(x)
    loss = uresnet_amd.train.compute_loss(out, y)
    loss.backward()
    res[tile] = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    res[tile]["__out"] = out.float()
names = list(res[True].keys())
for k in reversed(names):
    a, b = res[True][k].double(), res[False][k].double()
    print("%-32s %.3e   |g| %.3e" % (k, ((a - b).norm() / b.norm().clamp_min(1e-30)).item(), b.norm().item()))
