"""Report host synchronisations inside a training step (torch's sync debug mode warns at every blocking call)."""
import os, sys, warnings, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import uresnet_amd
from uresnet_amd import train
from uresnet_amd.modules.U_ResNet import UResNetNewEUCB
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = UResNetNewEUCB(19, 3, 64).set_compute_dtype(torch.bfloat16).to(dev).train()
opt = train.Adam(m.parameters(), lr=1e-4)
x = torch.randn(2, 3, 256, 512, device=dev)
y = torch.randint(0, 19, (2, 256, 512), device=dev)
for _ in range(3):
    train.seg_train_step(m, opt, x, y)
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    for _ in range(2):
        train.seg_train_step(m, opt, x, y)
torch.cuda.set_sync_debug_mode("default")
torch.cuda.synchronize()
print("synchronising calls in 2 steps:", len(w))
for i in w[:20]:
    print(str(i.message)[:200], i.filename, i.lineno)
print("adam table rebuilt?", opt._table_key is not None)
