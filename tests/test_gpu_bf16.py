"""The bf16-operand build (SELFC_OPERAND=bf16 -> libselfc_hip_bf16.so): same kernels, bfloat16 MFMA operands.
BASELINE.json names bf16 for the benchmark config; DESIGN.md section 2 explains why the default is f16 (same MFMA
rate, ~7x less rounding error).  This test measures both on the device: the bf16 path must work and land where
the CPU emulation predicted (a few 1e-3), i.e. outside the 1e-3 parity bar the f16 path meets."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

SCRIPT = r"""
import json, sys
sys.path.insert(0, %r)
import numpy as np, torch
from selfc_amd import GlobalVar, _lib
from selfc_amd.modules.SelfC_GMM_arch_inv import SelfCInvNet
GlobalVar.set_Temporal_LEN(7)
with np.load(%r) as z:
    g = {k: torch.from_numpy(np.asarray(z[k])) for k in z.files}
opt = {"global_module": "nonlocal", "stp_blk_num": 6, "fh_loss": "l2", "scale": 4, "gmm_k": 5}
net = SelfCInvNet(opt, 3, 3, "D2DTNet", [4, 4], 2)
net.load_state_dict({k: v for k, v in g.items() if k.startswith("operations.")}, strict=False)
net.cuda().eval()
rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
with torch.no_grad():
    zz, _ = net(x=g["x"].cuda(), rev=False)
    xr = net.inverse_from_latent(g["z"].cuda())
print(json.dumps({"operand": _lib.OPERAND, "lib": _lib.lib().selfc_version().decode(),
                  "fwd": rel(zz.cpu(), g["z"]), "inv": rel(xr.cpu(), g["x_rev"])}))
""" % (ROOT, os.path.join(ROOT, "tests", "golden", "g8_large_stack.npz"))


def _run(operand):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    env = dict(os.environ, SELFC_OPERAND=operand)
    env.pop("SELFC_LIB", None)
    p = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    return json.loads(p.stdout.strip().splitlines()[-1])


def test_bf16_operand_build_runs_and_is_less_accurate_than_f16():
    f = _run("f16")
    b = _run("bf16")
    assert f["operand"] == "f16" and "operands=f16" in f["lib"]
    assert b["operand"] == "bf16" and "operands=bf16" in b["lib"]
    assert f["fwd"] < 1e-3 and f["inv"] < 1e-3                  # the parity bar (BASELINE.json)
    assert 1e-3 < b["fwd"] < 1e-2 and 1e-3 < b["inv"] < 2e-2      # bf16: works, but outside the bar
    assert b["fwd"] > 3 * f["fwd"]
    print("f16:", f, "bf16:", b)


BWD_SCRIPT = r"""
import json, sys
sys.path.insert(0, %r)
sys.path.insert(0, %r)
import torch
from selfc_amd import GlobalVar, _lib
from selfc_amd.modules.Subnet_constructor import D2DTInput
from oracle import selfc_oracle as O
GlobalVar.set_Temporal_LEN(7)
torch.manual_seed(5)
m = D2DTInput(48, 3)
with torch.no_grad():
    for p in m.parameters():
        p.copy_(torch.randn_like(p) * (0.05 if p.dim() > 1 else 0.1))
sd = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
x = torch.randn(14, 48, 12, 20)
gy = torch.randn(14, 3, 12, 20) * 0.01
O.d2dt(sd, x, 7).backward(gy)
m.cuda()
m(x.cuda()).backward(gy.cuda())
l2 = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
print(json.dumps({"operand": _lib.OPERAND, "worst": max(l2(p.grad.cpu(), sd[n].grad) for n, p in m.named_parameters())}))
""" % (ROOT, os.path.join(ROOT, "tests"))


def test_backward_runs_on_both_operand_builds():
    """The gradient kernels are operand-type generic (scaled 16-bit planes, transposing LDS reads): bf16 works and is
    ~8x less accurate than f16 here as well."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    res = {}
    for operand in ("f16", "bf16"):
        env = dict(os.environ, SELFC_OPERAND=operand)
        env.pop("SELFC_LIB", None)
        p = subprocess.run([sys.executable, "-c", BWD_SCRIPT], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        res[operand] = json.loads(p.stdout.strip().splitlines()[-1])
    assert res["f16"]["worst"] < 3e-2
    assert res["bf16"]["worst"] < 2e-1 and res["bf16"]["operand"] == "bf16"
    print(res)
