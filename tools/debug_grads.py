import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import uresnet_amd
from oracle import uresnet_oracle as O
torch.manual_seed(0)
ref = O.UResNet(1, 3, 8).double()
g = torch.Generator().manual_seed(1234)
x = torch.randn(2, 3, 32, 32, generator=g)
y = (torch.rand(2, 1, 32, 32, generator=g) > 0.5).float()
m = uresnet_amd.modules.U_ResNet.UResNet(1, 3, 8).set_compute_dtype(torch.float32)
m.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
m.cuda().train()
names = ["Upsample", "Upsample.up_dwc", "Upsample.pwc", "ResLayer4", "ResLayer4.1", "ResLayer3", "up4", "out_conv", "Upsample.up_dwc.0", "Upsample.up_dwc.1", "Upsample.up_dwc.2"]
grads, outs = {}, {}
def hook(store_o, store_g, name):
    def f(mod, inp, out):
        store_o[name] = out.detach()
        out.register_hook(lambda gr: store_g.__setitem__(name, gr.detach()))
    return f
go, oo = {}, {}
for n in names:
    if n.count(".") < 2 or os.environ.get('UNFUSED'):
        dict(m.named_modules())[n].register_forward_hook(hook(outs, grads, n))
    dict(ref.named_modules())[n].register_forward_hook(hook(oo, go, n))
out = m(x.cuda())
uresnet_amd.train.bce_with_logits_loss(out, y.cuda()).backward()
O.bce_loss(ref(x.double()), y.double()).backward()
def host(t):
    return uresnet_amd.ops.to_nchw_raw(uresnet_amd.ops._g(t), torch.float32).cpu().double()
for n in names:
    if n in outs:
        eo = (host(outs[n]) - oo[n]).norm() / oo[n].norm()
        eg = (host(grads[n]) - go[n]).norm() / go[n].norm()
        print("%-20s out err %.2e   grad err %.2e  |g| %.2e" % (n, eo, eg, go[n].norm()))
if os.environ.get('UNFUSED'):
    zo = oo["Upsample.up_dwc"]            # post-ReLU output (oracle)
    zh = host(outs["Upsample.up_dwc"])
    mo, mh = zo > 0, zh > 0
    flips = (mo != mh)
    print("mask flips:", int(flips.sum()), "of", flips.numel())
    idx = flips.nonzero()[:10]
    for i in idx:
        i = tuple(i.tolist())
        print(i, "oracle z %.3e hip z %.3e  grad there %.3e" % (zo[i], zh[i], go["Upsample.up_dwc"][i]))
    # per-channel stats of the conv output feeding BN
    yo = oo["Upsample.up_dwc.1"]
    print("channels with zero variance:", int((yo.var((0, 2, 3)) == 0).sum()), " frac exact zeros in y: %.3f" % (yo == 0).double().mean())
    pre_o = oo["Upsample.up_dwc.2"]
