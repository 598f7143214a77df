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
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """tests/golden/<name>.npz -> {key: torch tensor} (fixtures made by tools/make_golden.py)."""
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: (torch.from_numpy(np.asarray(z[k])) if z[k].dtype.kind != "U" else [str(v) for v in z[k]]) for k in z.files}


def subdict(d, prefix):
    p = prefix + "."
    return {k[len(p):]: v for k, v in d.items() if k.startswith(p)}


@pytest.fixture(scope="session")
def golden():
    return load_golden


def rel_err(a, b):
    """max |a-b| / max |b|  (the "1e-3 relative fp32" metric of BASELINE.json, written down here)."""
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()


def seeded_fill(named_tensors, seed):
    """Deterministically fill parameters too large to store as fixtures (used identically by
    tools/make_golden.py on the reference module and by the tests on ours): names in sorted order,
    N(0, 1/fan_in) weights, N(0, 0.01^2) biases, drawn from one seeded CPU generator."""
    gen = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name in sorted(named_tensors):
            t = named_tensors[name]
            if t.dim() > 1:
                fan_in = int(np.prod(t.shape[1:]))
                t.copy_((torch.randn(t.shape, generator=gen) / fan_in ** 0.5).to(t.device))
            else:
                t.copy_((torch.randn(t.shape, generator=gen) * 0.01).to(t.device))
