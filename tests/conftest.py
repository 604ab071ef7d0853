import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "tiny-nerf-pytorch_amd")
for p in (ROOT, PKG, os.path.join(PKG, "src")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.device_count() > 0:          # counting devices does not initialise HIP in this process (is_available() does):
        return                                 # tests that start rank processes must be able to do so from a GPU-clean parent
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def load_golden(name):
    """Fixture arrays as torch tensors (numpy scalars stay python numbers)."""
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        out = {}
        for k in z.files:
            a = z[k]
            if a.dtype.kind in "US":
                out[k] = [str(s) for s in a.tolist()] if a.ndim else str(a)
            elif a.ndim == 0:
                out[k] = a.item()
            else:
                out[k] = torch.from_numpy(a.copy())
        return out


def golden_params(tag):
    g = load_golden(f"weights_{tag}")
    L, hidden, depth, skip_at = (int(v) for v in g["cfg"])
    params = [g[f"p{i:02d}"] for i in range(2 * depth + 4)]
    return dict(L=L, hidden=hidden, depth=depth, skip_at=skip_at, in_dim=6 * L + 3), params


@pytest.fixture(scope="session")
def golden():
    return load_golden
