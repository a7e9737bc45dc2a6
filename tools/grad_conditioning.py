"""Diagnostic: per-parameter gradient error of the HIP fp32 path and of the CPU fp32 oracle, both
measured against the fp64 oracle (shows which gradients are ill-conditioned by BN cancellation)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import uresnet_amd
from oracle import uresnet_oracle as O

name = sys.argv[1] if len(sys.argv) > 1 else "UResNet"
classes = int(sys.argv[2]) if len(sys.argv) > 2 else 1
shape = (2, 3, 32, 32)
torch.manual_seed(0)
ref = getattr(O, name)(classes, 3, 8)
g = torch.Generator().manual_seed(1234)
x = torch.randn(*shape, generator=g)
if classes > 1:
    y = torch.randint(0, classes, (shape[0], shape[2], shape[3]), generator=g)
    lossf = lambda o, d: O.compute_loss(o, y)
else:
    y = (torch.rand(shape[0], 1, shape[2], shape[3], generator=g) > 0.5).float()
    lossf = lambda o, d: O.bce_loss(o, y.to(d))
ref64 = getattr(O, name)(classes, 3, 8).double()
ref64.load_state_dict(ref.state_dict())
lossf(ref(x), torch.float32).backward()
lossf(ref64(x.double()), torch.float64).backward()
m = getattr(uresnet_amd.modules.U_ResNet, name)(classes, 3, 8).set_compute_dtype(torch.float32)
m.load_state_dict(ref.state_dict())
m.cuda().train()
out = m(x.cuda())
loss = uresnet_amd.train.compute_loss(out, y.cuda()) if classes > 1 else uresnet_amd.train.bce_with_logits_loss(out, y.cuda())
loss.backward()
rows = []
for (k, p), (_, q), (_, r) in zip(m.named_parameters(), ref.named_parameters(), ref64.named_parameters()):
    if r.grad is None:
        continue
    n = r.grad.norm().item()
    e_hip = (p.grad.cpu().double() - r.grad).norm().item() / max(n, 1e-30)
    e_cpu = (q.grad.double() - r.grad).norm().item() / max(n, 1e-30)
    rows.append((e_hip, e_cpu, n, k))
order = list(rows)
rows.sort(reverse=True)
print("%-34s %10s %10s %10s" % ("param", "hip32/64", "cpu32/64", "|g|"))
for e_hip, e_cpu, n, k in rows[:25]:
    print("%-34s %10.2e %10.2e %10.2e" % (k, e_hip, e_cpu, n))
import statistics
if os.environ.get('GC_ALL'):
    for e_hip, e_cpu, n, k in order:
        print('%-34s %10.2e %10.2e %10.2e' % (k, e_hip, e_cpu, n))
print("median hip %.2e cpu %.2e" % (statistics.median(r[0] for r in rows), statistics.median(r[1] for r in rows)))
