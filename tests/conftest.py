import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: z[k] for k in z.files}


def state_from(rec, prefix="w0/"):
    return {k[len(prefix):]: torch.from_numpy(v.copy()) for k, v in rec.items() if k.startswith(prefix)}


def rel_l2(a, b):
    a = torch.as_tensor(a).double().flatten()
    b = torch.as_tensor(b).double().flatten()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.fixture(scope="session")
def oracle():
    from oracle import uresnet_oracle
    return uresnet_oracle
