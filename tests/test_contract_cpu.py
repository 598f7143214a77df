"""Drop-in contract on CPU (no GPU needed): every mirrored net has exactly the reference's state_dict keys, tensor
shapes, parameter count and frozen parameters (tests/golden/state_dict_contract.json, written by
tools/make_golden.py from the reference modules), the init semantics the traps rely on hold, and the product
path refuses to run on CPU tensors instead of falling back."""
import json
import os

import pytest
import torch

from conftest import GOLDEN, ROOT
from selfc_amd import GlobalVar
from selfc_amd.modules import Inv_arch, SelfC_GMM_arch_inv, SelfC_arch_inv, Subnet_constructor as SC

CONTRACT = json.load(open(os.path.join(GOLDEN, "state_dict_contract.json")))
LARGE = {"global_module": "nonlocal", "stp_blk_num": 6, "scale": 4, "gmm_k": 5}
HAAR = {"stp_d2d_inner_c": 32, "stp_temporal_c": 32, "fh_loss": "l2", "gmm_mixture_num": 5, "stp_blk_num": 2}


def build(name):
    if name == "selfc_large_gmm":
        return SelfC_GMM_arch_inv.SelfCInvNet(dict(LARGE, fh_loss="gmm"), 3, 3, "D2DTNet", [4, 4], 2)
    if name == "selfc_large_l2":
        return SelfC_GMM_arch_inv.SelfCInvNet(dict(LARGE, fh_loss="l2"), 3, 3, "D2DTNet", [4, 4], 2)
    if name == "irn_dbnet":
        return Inv_arch.InvRescaleNet(3, 3, SC.subnet("DBNet", "xavier"), [2, 1], 2)
    if name == "selfc_haar_d2dt":
        return SelfC_arch_inv.SelfCInvNet(dict(HAAR, condition_func="D2DTNet"), 3, 3, "DBNet", [1], 1)
    return SelfC_arch_inv.SelfCInvNet(dict(HAAR, condition_func="FeatureCalapseBlock"), 3, 3, "DBNet", [1], 1)


@pytest.mark.parametrize("name", [k for k in CONTRACT if k != "init_facts"])
def test_state_dict_matches_reference(name):
    ref = CONTRACT[name]
    net = build(name)
    sd = net.state_dict()
    assert list(sd.keys()) == list(ref["shapes"].keys()) or sorted(sd.keys()) == sorted(ref["shapes"].keys())
    for k, shape in ref["shapes"].items():
        assert list(sd[k].shape) == shape, k
    assert sum(p.numel() for p in net.parameters()) == ref["n_params"]
    assert sorted(k for k, p in net.named_parameters() if not p.requires_grad) == ref["frozen"]
    # a reference checkpoint (optionally with DDP's "module." prefix stripped by base_model.load_network) loads strict
    net.load_state_dict({k: torch.zeros(s) for k, s in ref["shapes"].items()}, strict=True)


def test_init_semantics_trap4():
    facts = CONTRACT["init_facts"]
    torch.manual_seed(0)
    db = SC.DenseBlock(9, 3, "xavier")
    assert db.conv5.weight.abs().sum().item() == facts["denseblock_conv5_abs_sum"] == 0.0      # identity coupling at init
    assert sum(getattr(db, f"conv{i}").bias.abs().sum().item() for i in range(1, 6)) == facts["denseblock_bias_abs_sum"] == 0.0
    ratio = (db.conv1.weight.std() / (2.0 / (9 * 9 + 32 * 9)) ** 0.5).item()
    assert abs(ratio - 0.1) < 0.02 and abs(facts["denseblock_conv1_std_over_xavier"] - 0.1) < 0.02   # xavier_normal * 0.1
    d2 = SC.D2DTInput(48, 3, "xavier")
    assert facts["d2dt_conv5_is_nonzero"] and d2.conv5.weight.abs().sum() > 0    # Conv3d is skipped by the init helpers
    assert SC.subnet("NoSuchNet")(3, 3) is None                                   # unknown name -> None, as the reference


def test_no_cpu_fallback_anywhere():
    GlobalVar.set_Temporal_LEN(7)
    x = torch.zeros(7, 3, 16, 16)
    for name in ("selfc_large_l2", "irn_dbnet", "selfc_haar_d2dt"):
        with pytest.raises(RuntimeError, match="no CPU fallback|MI355X"):
            build(name)(x)
    with pytest.raises(RuntimeError):
        SelfC_GMM_arch_inv.FrequencyAnalyzer(3)(x)
    with pytest.raises(RuntimeError):
        Inv_arch.HaarDownsampling(3)(x)


def test_public_api_matches_reference():
    """Every mirrored class keeps the reference class's public methods and the parameter names / order / defaults of
    __init__ and forward (tests/golden/api_contract.json, dumped from the reference classes by tools/make_golden.py r2).
    Extra keyword parameters with defaults AFTER the reference's are allowed (e.g. FrequencyAnalyzer's k)."""
    import inspect
    from selfc_amd.modules import Quantization as Q, SelfC_Codec_arch_inv as CA
    from selfc_amd import global_var
    api = json.load(open(os.path.join(GOLDEN, "api_contract.json")))
    mods = {"Inv_arch": Inv_arch, "Subnet_constructor": SC, "SelfC_GMM_arch_inv": SelfC_GMM_arch_inv, "SelfC_arch_inv": SelfC_arch_inv,
            "SelfC_Codec_arch_inv": CA, "Quantization": Q, "global_var": global_var}

    def sig(f):
        return [[p.name, None if p.default is inspect._empty else repr(p.default)] for p in inspect.signature(f).parameters.values()]
    for name, ref in api.items():
        mod, cls_name = name.split(".")
        cls = getattr(mods[mod], cls_name)
        missing = [m for m in ref["methods"] if not hasattr(cls, m)]
        assert not missing, (name, missing)
        for what in ("init", "forward"):
            if ref[what] is None:
                continue
            ours = sig(cls.__init__ if what == "init" else cls.forward)
            want = [list(p) for p in ref[what]]
            assert ours[:len(want)] == want, (name, what, ours, want)
            assert all(d is not None for _, d in ours[len(want):]), (name, what, "extra parameters must have defaults")


def test_head_output_is_a_tensor_and_still_yields_module_parameters():
    """the STP nets publish the head output as ``self.parameters`` exactly as the reference does (a tensor that
    shadows nn.Module.parameters); here that tensor stays callable, so optimizers / zero_grad / deepcopy on the sub-module
    keep working after the first forward."""
    import copy
    import torch.nn as nn
    from selfc_amd.modules.module_util import HeadOutput
    m = nn.Linear(3, 2)
    y = m(torch.randn(4, 3))
    m.parameters = HeadOutput.wrap(y, m)
    assert torch.is_tensor(m.parameters) and m.parameters.shape == (4, 2)
    assert type(m.parameters * 2) is torch.Tensor and (m.parameters * 2).requires_grad      # plain results, graph intact
    assert [p.shape for p in m.parameters()] == [(2, 3), (2,)]
    m.zero_grad()
    torch.optim.Adam(m.parameters())
    (m.parameters ** 2).sum().backward()
    assert m.weight.grad is not None
    c = copy.deepcopy(m)                                                                    # non-leaf attribute: copied detached
    assert type(c.parameters) is torch.Tensor and not c.parameters.requires_grad


def test_module_with_published_head_output_pickles():
    """After its first forward an STP net carries `parameters` = HeadOutput (a non-leaf tensor with a weak reference to the module):
    torch.save(net) / a spawn hand-off must still work - the head output goes over as a detached plain tensor."""
    import io
    import pickle
    import torch
    import torch.nn as nn
    from selfc_amd.modules.module_util import HeadOutput
    m = nn.Linear(2, 2)
    m.parameters = HeadOutput.wrap(torch.randn(3, 2) @ m.weight, m)
    buf = io.BytesIO()
    torch.save(m, buf)
    buf.seek(0)
    m2 = torch.load(buf, weights_only=False)
    got = m2.__dict__["parameters"]
    assert type(got) is torch.Tensor and not got.requires_grad and torch.equal(got, m.parameters.detach())
    assert pickle.loads(pickle.dumps(m.parameters)).shape == (3, 2)


def test_bench_host_weights_checks_its_copies(tmp_path, monkeypatch):
    """bench.py feeds the CPU oracle a host copy of the weights that it verifies against the source tensor (a reference computed
    from a wrong copy reads as a parity failure of the kernels: profiles/r4/parity_leg_host_copy.txt).  On host tensors the
    check is the identity: every `operations.*` tensor comes back equal, no error is recorded; other keys are not taken.  The
    per-tensor hash must move with what a sum of |w| cannot see (a swapped pair, a sign flip), and a wrong copy must be described
    (where, what, what else holds those bytes) rather than silently replaced."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.operations = torch.nn.ModuleList([torch.nn.Conv2d(3, 4, 3), torch.nn.Conv2d(4, 2, 1)])
            self.stp_net = torch.nn.Linear(2, 2)

    net = Net()
    params, errors = bench.host_weights(net)
    assert errors == []
    assert set(params) == {k for k in net.state_dict() if k.startswith("operations.")} and len(params) == 4
    assert all(torch.equal(params[k], net.state_dict()[k]) for k in params)
    h0 = bench.weights_hash(net)
    assert h0 == bench.weights_hash(net) and len(h0) == 6
    w = net.operations[0].weight.detach()
    sw = w.clone()
    sw.view(-1)[[0, 1]] = w.view(-1)[[1, 0]]
    assert bench.tensor_hash(sw) != bench.tensor_hash(w) and bench.tensor_hash(-w) != bench.tensor_hash(w)
    assert float(sw.abs().sum()) == float(w.abs().sum())             # what the old checksum saw: nothing
    # a wrong copy: bytes 64..127 replaced by the bytes of another buffer
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    other = torch.randn_like(w)
    wrong = w.clone()
    wrong.view(-1)[16:32] = other.view(-1)[16:32]
    info = bench.describe_bad_copy("operations.0.weight", wrong, w, {"other": other, "unrelated": torch.zeros_like(w)})
    assert info["first_bad_byte"] >= 64 and info["last_bad_byte"] <= 127 and info["bad_bytes"] >= 56     # (a byte may coincide)
    assert all(64 <= a < b <= 128 for a, b in info["bad_runs"])
    assert info["wrong_bytes_equal_the_same_region_of"] == ["other"] and not info["wrong_bytes_all_zero"]
    assert info["second_copy_matches_the_device"] and os.path.exists(os.path.join(str(tmp_path), info["dump"]))
    assert "host" in bench.box_identity()


def test_bench_compact_line_is_what_the_driver_keeps():
    """bench.py prints ONE line of the driver's standard keys + config / roofline / cpu_baseline with scalar values only and
    strings under 120 characters (the driver's record drops nested objects and every other top-level key: BENCH_r05 lost all
    secondary legs that way); the secondary legs travel as flat scalars inside `roofline`."""
    import json
    import bench
    out = {"metric": "m", "value": 1500.0, "unit": "septuplets/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 2.6, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic", "rccl_ranks": 1,
           "config": {"workload": "w" * 300, "septuplets_per_gpu": 4, "launch": "hipGraph replay (one graph)", "streams": 2, "sharding": "s" * 200, "prewarm": "p" * 200,
                      "graph_form_probe": {"a": 1}},
           "box_calibration": {"shader_clock_GHz_under_the_workload": 2.4, "mfma_f16_loop_TFLOPs": 2500.0, "device_copy_GBps": 5000.0},
           "box_identity": {"host": "h"},
           "roofline": {"bound": "mfma", "kernel": "k" * 200, "achieved": 660.0, "peak": 2500.0, "unit": "TFLOP/s", "frac": 0.264, "traffic": 1.0e8,
                        "avg_launch_us": 71.0, "rocprof_avg_us": 77.0, "pmc_reference_file": "profiles/r6/pmc_traffic.json"},
           "roofline_other": {"conv5_GH": {"achieved": 4500.0, "frac": 0.56}, "conv3x3": {"frac": 0.3}},
           "stack_roofline": {"mfma_frac": 0.26, "hbm_frac_layer_granular": 0.55},
           "train_step": {"ms_per_step": 11.0, "captured_ms_per_step_by_local_batch": {"1": 6.0, "2": 7.0, "4": 8.0, "8": 11.0}, "mfma_frac": 0.08,
                          "graph_nodes": 600, "graph_nodes_b1": 590, "eager_ms_per_step": 17.0, "pmc_reference_file": {"file": "x", "same_sources": None}},
           "uvg_1080p": {"frames_per_s": 430.0, "mfma_frac_whole_path": 0.24}, "full_test_path": {"septuplets_per_s": 1090.0, "mfma_frac_whole_path": 0.237},
           "headline_through_module_api": {"septuplets_per_s": 1440.0},
           "parity": {"fwd_latent_rel_err": 3.2e-4, "inv_rel_err": 5.9e-4, "fwd_latent_rel_l2": 2.8e-4, "inv_rel_l2": 2.9e-4},
           "cpu_baseline": {"value": 1.3, "unit": "septuplets/s", "cores": 16, "kind": "port", "cpu_model": "EPYC", "host_copy_errors": [], "sample": "x" * 300}}
    line = bench.compact_line(out)
    allowed = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
               "config", "roofline", "cpu_baseline"}
    assert set(line) == allowed
    for key in ("config", "roofline", "cpu_baseline"):
        for k, v in line[key].items():
            assert not isinstance(v, (dict, list)), (key, k)
            assert not isinstance(v, str) or len(v) < 120, (key, k, len(v))
    r = line["roofline"]
    assert r["train_step_ms_b8"] == 11.0 and r["train_step_ms_b1"] == 6.0 and r["train_step_graph_nodes"] == 600 and r["uvg_1080p_frames_per_s"] == 430.0
    assert r["full_test_path_septuplets_per_s"] == 1090.0 and r["parity_inv_rel_err"] == 5.9e-4 and r["rocprof_avg_us"] == 77.0
    assert len(json.dumps(line)) < 3600


def test_shutdown_marks_the_package_instead_of_leaving_dead_streams_behind():
    """runtime._shutdown (atexit) destroys the package's own HIP streams; an atexit handler registered before the import
    runs after it.  From then on the module API must take its eager path (pipeline.module_graph_call checks rt.SHUT_DOWN) and a
    multi-stream pipeline must refuse loudly."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from selfc_amd import runtime as rt, pipeline\n"
            "assert rt.SHUT_DOWN is False\n"
            "rt._shutdown()\n"
            "assert rt.SHUT_DOWN is True\n"
            "import inspect\n"
            "assert 'rt.SHUT_DOWN' in inspect.getsource(pipeline)\n"
            "m = pipeline.MultiStreamRoundTrip.__new__(pipeline.MultiStreamRoundTrip)\n"
            "try:\n    m.replay()\nexcept RuntimeError as e:\n    assert 'shut down' in str(e); print('refused')\n" % ROOT)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "refused" in p.stdout, p.stderr[-1500:]


def test_parameter_list_cache_follows_every_registration_and_costs_nothing_in_between():
    """runtime.plist: one list object per parameter set (callers key caches on its identity); a parameter replaced ANYWHERE in the
    module gives a new list, a registration in an unrelated module does not, and between registrations the call does not
    walk the module tree (it was 0.7 ms per whole-net call on the module-API path)."""
    import time
    from selfc_amd import runtime as rt
    GlobalVar.set_Temporal_LEN(7)
    blk = Inv_arch.InvBlockExp(lambda cin, cout: SC.subnet("D2DTNet")(cin, cout), 51, 3)
    l0 = rt.plist(blk)
    assert rt.plist(blk) is l0 and [id(p) for p in l0] == [id(p) for p in blk.parameters()]
    t0 = time.perf_counter()
    for _ in range(1000):
        rt.plist(blk)
    assert (time.perf_counter() - t0) / 1000 < 20e-6
    blk.H.conv4.bias = torch.nn.Parameter(blk.H.conv4.bias.detach().clone())          # not the first parameter
    l1 = rt.plist(blk)
    assert l1 is not l0 and [id(p) for p in l1] == [id(p) for p in blk.parameters()]
    torch.nn.Linear(2, 2)                                                               # a registration elsewhere
    assert rt.plist(blk) is l1
    blk.float()                                                                         # _apply keeps the Parameter objects
    assert rt.plist(blk) is l1
