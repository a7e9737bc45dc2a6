import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import uresnet_amd
from uresnet_amd import ops, _lib
from uresnet_amd._lib import call, query
torch.manual_seed(0)
n,c,h,w = 2,32,8,8
rows = n*h*w
y = ops.to_nhwc_raw(torch.randn(n,c,h,w).cuda(), torch.float32)
dz = ops.to_nhwc_raw(torch.randn(n,c,h,w).cuda(), torch.float32)
mean = torch.zeros(c, device='cuda'); invstd = torch.ones(c, device='cuda'); gamma = torch.ones(c, device='cuda'); beta = torch.zeros(c, device='cuda')
nparts = query("ur_bn_stats_rows", rows, c)
partial = torch.full((nparts, c, 2), -7.0, device='cuda')
for z, act in ((None, 1), (y, 1), (None, 0)):
    call("ur_bn_act_bwd_reduce", dz.data_ptr(), y.data_ptr(), None if z is None else z.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), act, 0, rows, c, partial.data_ptr(), ops._stream())
    torch.cuda.synchronize()
    print("z" if z is not None else "no z", act, nparts, partial.sum(0)[:4].cpu().tolist())
yy = y.permute(0,2,3,1).reshape(-1,c); dd = dz.permute(0,2,3,1).reshape(-1,c)
g = dd*(yy>0)
print("want", torch.stack([g.sum(0), (g*yy).sum(0)],1)[:4].cpu().tolist())

print("want act0", torch.stack([dd.sum(0), (dd*yy).sum(0)],1)[:2].cpu().tolist())
st = torch.full((query("ur_bn_stats_rows", rows, c), c, 2), -7.0, device='cuda')
call("ur_bn_stats", y.data_ptr(), 0, rows, c, st.data_ptr(), ops._stream()); torch.cuda.synchronize()
print("stats", st.sum(0)[:2].cpu().tolist(), "want", torch.stack([yy.sum(0), (yy*yy).sum(0)],1)[:2].cpu().tolist())
