import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import uresnet_amd
from uresnet_amd import ops, nn as hnn
torch.manual_seed(0)
def rel(a, b): return ((a.double()-b.double()).norm()/b.double().norm()).item()
def host(t): return ops.to_nchw_raw(ops._g(t.detach()), torch.float32).cpu()
for (n,c,h,w) in [(2,32,32,32),(2,32,16,32),(1,64,8,8),(2,80,16,32)]:
    x = torch.randn(n,c,h,w)*2+0.5
    gy = torch.randn(n,c,h,w)
    bn = torch.nn.BatchNorm2d(c).double(); bn.weight.data.uniform_(0.5,1.5); bn.bias.data.normal_()
    xr = x.double().requires_grad_(True)
    zr = F.relu(bn(xr)); (zr*gy.double()).sum().backward()
    b2 = hnn.BatchNorm2d(c); b2.load_state_dict({k: v.float() if v.is_floating_point() else v for k,v in torch.nn.BatchNorm2d(c).state_dict().items()})
    b2.weight.data.copy_(bn.weight.data.float()); b2.bias.data.copy_(bn.bias.data.float()); b2.cuda()
    xd = x.cuda().requires_grad_(True)
    z = ops.bn_act(ops.as_activation(xd, torch.float32), b2, act=1)
    (z*ops.to_nhwc_raw(gy.cuda(), torch.float32)).sum().backward()
    print((n,c,h,w), "z %.1e dx %.1e dgamma %.1e dbeta %.1e" % (rel(host(z), zr), rel(xd.grad.cpu(), xr.grad), rel(b2.weight.grad.cpu(), bn.weight.grad), rel(b2.bias.grad.cpu(), bn.bias.grad)))
# chain: upsample -> depthwise -> bn relu
x = torch.randn(2,32,16,16); wt = torch.randn(32,1,3,3)/3; gy = torch.randn(2,32,32,32)
xr = x.double().requires_grad_(True); wr = wt.double().requires_grad_(True)
u = F.interpolate(xr, scale_factor=2, mode="bilinear"); u.retain_grad()
yr = F.conv2d(u, wr, padding=1, groups=32); yr.retain_grad()
(yr*gy.double()).sum().backward()
xd = x.cuda().requires_grad_(True); wd = wt.cuda().requires_grad_(True)
ud = ops.upsample_bilinear(ops.as_activation(xd, torch.float32), 2); ud.retain_grad()
yd = ops.gconv3x3(ud, wd)
(yd*ops.to_nhwc_raw(gy.cuda(), torch.float32)).sum().backward()
print("chain y %.1e du %.1e dx %.1e dw %.1e" % (rel(host(yd), yr), rel(host(ud.grad), u.grad), rel(xd.grad.cpu(), xr.grad), rel(wd.grad.cpu(), wr.grad)))
