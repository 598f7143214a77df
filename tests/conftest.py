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
    # the oracle is torch on the CPU: on the GPU box's 256-thread host its convs collapse with the default thread count (bench.py
    # calibrates for the same reason: 355 s per septuplet at 256 threads against 0.8 s at 16)
    torch.set_num_threads(min(16, os.cpu_count() or 16))


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


#: every parity figure the tests measured: written to gpurun_out/parity_report.json at session end (and copied to
#: profiles/rN/ by hand), so a bound in an assert can always be read next to the value that was measured under it
_PARITY_LOG = []


def _record(metric, value):
    import inspect
    fr = inspect.currentframe()
    while fr is not None and os.path.basename(fr.f_code.co_filename) == "conftest.py":
        fr = fr.f_back
    where = f"{os.path.basename(fr.f_code.co_filename)}:{fr.f_lineno}" if fr is not None else "?"
    _PARITY_LOG.append({"test": os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0], "metric": metric,
                        "where": where, "value": value})
    return value


def record(metric, value):
    """log any other measured figure (e.g. gradient-norm ratios) next to the parity values"""
    return _record(metric, float(value))


def rel_err(a, b):
    """max |a-b| / max |b|  (the "1e-3 relative fp32" metric of BASELINE.json, written down here)."""
    return _record("max|a-b|/max|b|", ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item())


def rel_l2(a, b):
    """||a-b||_2 / ||b||_2: the two-sided companion of rel_err (not dominated by the largest element)."""
    return _record("||a-b||/||b||", ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item())


def group_err(a, b, split=3):
    """(rel_err, rel_l2) of the LR channels [0, split) and of the HF channels [split, C) SEPARATELY: on a 51-channel latent
    the HF channels must not be judged against the LR channels' magnitude.  Returns the worst of the four figures."""
    worst = 0.0
    for sl in (slice(0, split), slice(split, None)):
        worst = max(worst, rel_err(a[:, sl], b[:, sl]), rel_l2(a[:, sl], b[:, sl]))
    return worst


def pytest_sessionfinish(session, exitstatus):
    if not _PARITY_LOG:
        return
    import json
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        gpu = any("test_gpu" in e["test"] for e in _PARITY_LOG)
        with open(os.path.join(out, "parity_report_gpu.json" if gpu else "parity_report_cpu.json"), "w") as fh:
            json.dump(_PARITY_LOG, fh, indent=0)
    except OSError:
        pass


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
