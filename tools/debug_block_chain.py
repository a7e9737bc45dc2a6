"""Operator-by-operator teacher-forced comparison inside one residual block (bf16): every HIP operator of
BasicBlock(C, C) is fed the bf16-storage oracle's own tensors and compared with the oracle's result for that operator
alone, forward and backward.  Prints one line per operator; shows where a block-level difference comes from."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import uresnet_amd                                   # noqa: E402
from uresnet_amd import _lib, ops                    # noqa: E402
from uresnet_amd import nn as hnn                    # noqa: E402
from oracle import uresnet_oracle as O               # noqa: E402

DEV = "cuda:0"
C, H, W = int(os.environ.get("C", 64)), int(os.environ.get("H", 256)), int(os.environ.get("W", 512))


def rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def frac_diff(a, b):
    return (a != b).double().mean().item()


def host(t):
    return ops.to_nchw_raw(ops._g(t.detach()), torch.float32).cpu()


def dev(t):
    return ops.as_activation(t.to(DEV), torch.bfloat16).detach()


torch.manual_seed(0)
blk = O.BasicBlock(C, C).train()
O.emulate_bf16_storage(blk)
g = torch.Generator().manual_seed(1)
x = O._bf16(torch.relu(torch.randn(1, C, H, W, generator=g)))
gout = O._bf16(torch.randn(1, C, H, W, generator=g) * 1e-3)
rec = {}
mods = {"conv1": blk.left[0], "bn1": blk.left[1], "conv2": blk.left[3], "bn2": blk.left[4]}
for name, m in mods.items():
    def post(mod, args, out, name=name):
        rec[name + ".in"] = args[0].detach().clone()
        rec[name + ".out"] = out.detach().clone()
        out.register_hook(lambda gr, name=name: rec.__setitem__(name + ".gout", gr.detach().clone()))
        args[0].requires_grad and args[0].register_hook(lambda gr, name=name: rec.__setitem__(name + ".gin", gr.detach().clone()))
    m.register_forward_hook(post)
xr = x.clone().requires_grad_(True)
outr = blk(xr)
outr.backward(gout)
z1r = torch.relu(rec["bn1.out"])                      # what conv2 consumed
print("shapes", tuple(x.shape), "| oracle tensors recorded:", sorted(rec))

# ---- forward, operator by operator
w1, w2 = blk.left[0].weight.detach().to(DEV), blk.left[3].weight.detach().to(DEV)
y1, st1 = ops.conv2d(dev(x), w1, None, ops.PackedWeights(), 1, 1, want_stats=True)
print("conv1 fwd      rel %.2e  differing elements %.4f" % (rel(host(y1), rec["conv1.out"]), frac_diff(host(y1), rec["conv1.out"])))
bn1 = hnn.BatchNorm2d(C).to(DEV).train()
z1 = ops.bn_act(dev(rec["conv1.out"]), bn1)
print("bn1+relu fwd   rel %.2e  differing %.4f   (stats from the stored tensor)" % (rel(host(z1), z1r), frac_diff(host(z1), z1r)))
z1b = ops.bn_act(y1, hnn.BatchNorm2d(C).to(DEV).train(), stats=st1)
print("bn1+relu fwd   rel %.2e  differing %.4f   (HIP conv output, statistics from the fp32 accumulators)" % (rel(host(z1b), z1r), frac_diff(host(z1b), z1r)))
y2, st2 = ops.conv2d(dev(z1r), w2, None, ops.PackedWeights(), 1, 1, want_stats=True)
print("conv2 fwd      rel %.2e  differing %.4f" % (rel(host(y2), rec["conv2.out"]), frac_diff(host(y2), rec["conv2.out"])))
bn2 = hnn.BatchNorm2d(C).to(DEV).train()
y2d = dev(rec["conv2.out"]).requires_grad_(True)
resd = dev(x).requires_grad_(True)
out = ops.bn_act(y2d, bn2, residual=resd)
print("bn2+res+relu   rel %.2e  differing %.4f" % (rel(host(out), outr.detach()), frac_diff(host(out), outr.detach())))

# ---- backward, operator by operator (each fed the oracle's incoming gradient, rounded as stored)
out.backward(dev(gout))
want_dy2 = O._bf16(rec["conv2.gout"])
print("bn2 bwd dy     rel %.2e  differing %.4f" % (rel(host(y2d.grad), want_dy2), frac_diff(host(y2d.grad), want_dy2)))
print("bn2 bwd dgamma rel %.2e  dbeta %.2e" % (rel(bn2.weight.grad.cpu(), blk.left[4].weight.grad), rel(bn2.bias.grad.cpu(), blk.left[4].bias.grad)))
# conv2 backward
z1d = dev(z1r).requires_grad_(True)
w2p = w2.clone().requires_grad_(True)
y2b, _ = ops.conv2d(z1d, w2p, None, ops.PackedWeights(), 1, 1)
y2b.backward(dev(want_dy2))
want_dz1 = rec["conv2.gin"]
print("conv2 dgrad    rel %.2e   (vs unrounded oracle gradient; bf16 storage alone gives about 1.6e-3)" % rel(host(z1d.grad), want_dz1))
print("conv2 wgrad    rel %.2e" % rel(w2p.grad.cpu(), blk.left[3].weight.grad))
# bn1 backward
y1d = dev(rec["conv1.out"]).requires_grad_(True)
bn1b = hnn.BatchNorm2d(C).to(DEV).train()
z1c = ops.bn_act(y1d, bn1b)
z1c.backward(dev(O._bf16(want_dz1)))
want_dy1 = O._bf16(rec["conv1.gout"])
print("bn1 bwd dy     rel %.2e  differing %.4f" % (rel(host(y1d.grad), want_dy1), frac_diff(host(y1d.grad), want_dy1)))
print("bn1 bwd dgamma rel %.2e  dbeta %.2e" % (rel(bn1b.weight.grad.cpu(), blk.left[1].weight.grad), rel(bn1b.bias.grad.cpu(), blk.left[1].bias.grad)))
xd = dev(x).requires_grad_(True)
w1p = w1.clone().requires_grad_(True)
y1b, _ = ops.conv2d(xd, w1p, None, ops.PackedWeights(), 1, 1)
y1b.backward(dev(want_dy1))
print("conv1 dgrad    rel %.2e" % rel(host(xd.grad), rec["conv1.gin"] - 0 if "conv1.gin" in rec else xr.grad - gout * (outr.detach() > 0)))
print("conv1 wgrad    rel %.2e" % rel(w1p.grad.cpu(), blk.left[0].weight.grad))
