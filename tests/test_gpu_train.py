"""One optimisation step through the HIP path (selfc_amd/train.py: the restated SelfCModel.optimize_parameters) against
the G11 fixture: the same step run on the reference modules with stock autograd (tools/make_golden.py).  Losses are
forward quantities (1e-3); gradients see f16 operands in both directions plus LeakyReLU kink flips (see
test_gpu_backward.py), so their bars are relative L2 errors."""
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
T = 7
OPT = {"global_module": "nonlocal", "stp_blk_num": 6, "fh_loss": "l2", "scale": 4, "gmm_k": 5}


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from selfc_amd import _lib, GlobalVar
    _lib.lib()
    GlobalVar.set_Temporal_LEN(T)
    return torch.device("cuda:0")


def _net(dev, fh_loss="l2"):
    from selfc_amd.modules.SelfC_GMM_arch_inv import SelfCInvNet
    net = SelfCInvNet(dict(OPT, fh_loss=fh_loss), 3, 3, "D2DTNet", [4, 4], 2)
    sd = {k: v for k, v in load_golden("g8_large_stack").items() if k.startswith("operations.")}
    if fh_loss == "l2":
        sd.update({k: v for k, v in load_golden("g7_stp_l2_full_rev").items() if k.startswith("stp_net.")})
        net.load_state_dict(sd, strict=True)
    else:
        net.load_state_dict(sd, strict=False)
    return net.to(dev)


def test_train_step_matches_reference_step(dev):
    from selfc_amd import train
    g = load_golden("g11_train_step")
    x = load_golden("g8_large_stack")["x"]                           # (T,3,32,48) = one clip
    net = _net(dev)
    before = {n: p.detach().clone() for n, p in net.named_parameters()}
    tr = train.RescaleTrainer(net, dict(train.TRAIN_OPT_LARGE))
    gt = x.reshape(1, T, 3, 32, 48).transpose(1, 2).to(dev)          # data['GT'] (B,C,T,H,W)
    real_h, ref_l, clip_len = train.feed_data(gt, "sr_bd", 4)
    assert clip_len == T and torch.equal(real_h.cpu(), x)
    assert float((ref_l.cpu() - g["ref_l"]).abs().max()) < 1e-5
    # keep the unclipped gradients: the clip rescales .grad in place.  (The trainer's gradients live in one flat buffer
    # the weight-gradient kernels add into - autograd.GradSink - so they are read there, not through tensor hooks.)
    assert tr.sink is not None
    grads = {}
    tr.before_clip = lambda t_: grads.update({n: p.grad.detach().clone() for n, p in net.named_parameters()})
    log = tr.optimize_parameters(real_h, ref_l)
    assert abs(log["l_forw_fit"] - float(g["l_forw_fit"])) < 1e-3 * float(g["l_forw_fit"])
    assert abs(log["l_back_rec"] - float(g["l_back_rec"])) < 1e-3 * float(g["l_back_rec"])
    assert abs(log["loss"] - float(g["loss"])) < 1e-3 * float(g["loss"])
    names = g["names"]
    norms = torch.tensor([float(grads[n].norm()) for n in names], dtype=torch.float64)
    ref = g["grad_norms"]
    from conftest import record
    rel = record("||grad norms - ref|| / ||ref|| over the 350 tensors", (norms - ref).norm() / ref.norm())
    ratio = record("median per-tensor gradient-norm ratio", (norms / ref).median())
    total = record("total gradient norm ratio", float(tr.grad_norm) / float(g["grad_norm"]))
    # measured on the MI355X: 1.7e-4, 0.99991, 0.99989 (profiles/r2/parity_report_gpu.json); bars 10x above that - far
    # below the 1 % a systematic scale error of the weight-gradient path (e.g. a wrong 1/S) would cause
    assert rel < 2e-3, rel
    assert abs(ratio - 1.0) < 2e-3 and abs(total - 1.0) < 2e-3, (ratio, total)
    # one clipped gradient tensor element-wise, and the Adam step itself: |delta| <= lr, direction = -sign(grad)
    clipped = net.operations[1].F.conv1.weight.grad.cpu()
    e = float((clipped - g["grad_F1_conv1_clipped"]).norm() / g["grad_F1_conv1_clipped"].norm())
    assert e < 5e-2, e
    for n, p in net.named_parameters():
        d = (p.detach() - before[n]).cpu()
        assert float(d.abs().max()) <= 1e-4 * 1.001, n
    d = (net.operations[1].F.conv1.weight.detach() - before["operations.1.F.conv1.weight"]).cpu()
    big = g["grad_F1_conv1_clipped"].abs() > 1e-3 * g["grad_F1_conv1_clipped"].abs().max()
    agree = (torch.sign(d[big]) == -torch.sign(g["grad_F1_conv1_clipped"][big])).float().mean()
    assert float(agree) > 0.98


@pytest.mark.parametrize("mode", ["eager", "captured"])
def test_k20_training_trajectory_tracks_the_reference(dev, mode):
    """K = 20 consecutive optimisation steps on one fixed 32x48 clip against fixture g18 (tools/make_golden_r6.py: the SAME 20
    steps on the imported reference modules, fp32 CPU, stock autograd, torch Adam; SelfC_model.py:148-183): every step's
    l_forw_fit / l_back_rec within 1 % of the reference trajectory's, and the accumulated weight update (final - initial over
    the 350 trainable tensors) compared as ONE vector.  The f16-operand steps do not drift: each step's losses are forward
    quantities of the weights the previous 1..k-1 f16-gradient steps produced."""
    from conftest import record
    from selfc_amd import train
    g = load_golden("g18_train_trajectory")
    x = load_golden("g8_large_stack")["x"]
    net = _net(dev)
    names = [n for n, p in net.named_parameters() if p.requires_grad]
    assert names == g["names"]
    before = [p.detach().clone() for n, p in net.named_parameters() if p.requires_grad]
    cap = mode == "captured"
    tr = train.RescaleTrainer(net, dict(train.TRAIN_OPT_LARGE), capturable=cap)
    gt = x.reshape(1, T, 3, 32, 48).transpose(1, 2).to(dev)
    real_h, ref_l, _ = train.feed_data(gt, "sr_bd", 4)
    K = len(g["loss"])
    logs = []
    if cap:
        tr.capture(real_h, ref_l, warmup=3)                         # 3 real (eager, capturable-Adam) steps, then replays
        logs += tr.warmup_logs
    while len(logs) < K:
        logs.append(dict(tr.optimize_parameters(real_h, ref_l)))
    worst_fit = max(abs(l["l_forw_fit"] - float(r)) / float(r) for l, r in zip(logs, g["l_forw_fit"]))
    worst_rec = max(abs(l["l_back_rec"] - float(r)) / float(r) for l, r in zip(logs, g["l_back_rec"]))
    worst_gn = max(abs(l["grad_norm"] - float(r)) / float(r) for l, r in zip(logs, g["grad_norm"]))
    record(f"K=20 trajectory ({mode}): worst per-step l_forw_fit relative error", worst_fit)
    record(f"K=20 trajectory ({mode}): worst per-step l_back_rec relative error", worst_rec)
    record(f"K=20 trajectory ({mode}): worst per-step total-gradient-norm relative error", worst_gn)
    assert worst_fit < 1e-2 and worst_rec < 1e-2, (worst_fit, worst_rec)
    # the update vector.  Adam's per-element step is lr * m / (sqrt(v) + eps): elements whose gradient is noise-sized (sign flips from
    # one step to the next) move by +-lr whatever the size of the gradient, so the vector's relative L2 error is far above a gradient's;
    # measured on the MI355X (profiles/r6/parity_report_gpu.json): losses 7e-4 / 9e-5, gradient norm 1.1e-2 worst step, update 1.1e-2
    ref = g["update_q"].float() * float(g["update_scale"])
    after = [p.detach() for n, p in net.named_parameters() if p.requires_grad]
    upd = torch.cat([(a - b).flatten() for a, b in zip(after, before)]).cpu()
    e_all = record(f"K=20 trajectory ({mode}): ||update - ref|| / ||ref|| over all 3.37 M weights", (upd - ref).norm() / ref.norm())
    cos = record(f"K=20 trajectory ({mode}): cosine(update, ref)", torch.dot(upd, ref) / (upd.norm() * ref.norm()))
    record(f"K=20 trajectory ({mode}): ||update|| / ||ref||", upd.norm() / ref.norm())
    assert e_all < 0.035 and cos > 0.999, (e_all, cos)          # measured 1.08e-2 / 1.04e-2 (eager / captured), cosine 1.0000


def test_flat_gradient_sink_equals_autograd_accumulation(dev):
    """RescaleTrainer(flat_grads=True): gradients accumulated by the kernels into one buffer (beta = 1) and clipped there -
    the same step as stock autograd accumulation + clip_grad_norm_ (two calls per block and step: the sums differ only in
    the order of two fp32 additions).  Two steps (the second checks the re-zeroing of the buffer); the two trainers start the
    second step from the SAME weights: their clip coefficients differ in the last bit (a staged norm of one buffer against
    clip_grad_norm_ over 350 tensors), and a 1e-8 weight difference moves the f16 kernels' next gradients by up to 0.5 % (LeakyReLU
    kinks) - equivalence of the two accumulation paths is a statement about equal weights."""
    from selfc_amd import train
    x = load_golden("g8_large_stack")["x"]
    gt = x.reshape(1, T, 3, 32, 48).transpose(1, 2).to(dev)
    real_h, ref_l, _ = train.feed_data(gt, "sr_bd", 4)
    nets, trs, got = {}, {}, {}
    for flat in (False, True):
        nets[flat] = _net(dev)
        trs[flat] = train.RescaleTrainer(nets[flat], dict(train.TRAIN_OPT_LARGE), flat_grads=flat)
        assert (trs[flat].sink is not None) == flat
        got[flat] = {}
        trs[flat].before_clip = lambda t_, net=nets[flat], g_=got[flat]: g_.update({n: p.grad.detach().clone() for n, p in net.named_parameters()})
    from conftest import record
    for step in range(2):
        logs = {flat: trs[flat].optimize_parameters(real_h, ref_l)["loss"] for flat in (False, True)}
        ga, gb = got[False], got[True]
        assert set(ga) == set(gb) and all(v is not None for v in gb.values())
        # per-tensor relative L2, against a floor of 1e-7 of the whole gradient's norm: the proj3 biases of the GlobalAgg blocks have
        # a gradient that is ZERO in exact arithmetic (a bias in front of a softmax over its own axis shifts every logit alike) - what
        # the kernels deliver for them is cancellation noise at 1e-13 of the gradient's norm
        g_all = float(torch.sqrt(sum((v.double() ** 2).sum() for v in ga.values())))
        worst = max(float((ga[n] - gb[n]).norm() / (ga[n].norm() + 1e-7 * g_all)) for n in ga)
        record(f"flat gradient sink vs autograd accumulation, worst per-tensor relative L2 (step {step + 1})", worst)
        assert worst < 1e-4, (step, worst)
        na, nb = float(trs[False].grad_norm), float(trs[True].grad_norm)
        assert abs(na - nb) < 1e-5 * na and abs(logs[False] - logs[True]) < 1e-5 * abs(logs[False])
        pd = max(float((a - b).abs().max()) for a, b in zip(nets[False].state_dict().values(), nets[True].state_dict().values()))
        assert pd < 2e-6, (step, pd)                            # the same update
        if step == 0:                                           # same weights into the second step (docstring)
            with torch.no_grad():
                for a, b in zip(nets[False].state_dict().values(), nets[True].state_dict().values()):
                    b.copy_(a)
            from selfc_amd import runtime as rt
            rt.invalidate_weights()


def test_gmm_training_reduces_loss(dev):
    """The shipped configuration (fh_loss: gmm, device RNG): a few steps on one fixed clip must reduce the loss."""
    from selfc_amd import train
    torch.manual_seed(0)
    net = _net(dev, "gmm")
    tr = train.RescaleTrainer(net, dict(train.TRAIN_OPT_LARGE))
    x = load_golden("g8_large_stack")["x"]
    gt = x.reshape(1, T, 3, 32, 48).transpose(1, 2).to(dev)
    real_h, ref_l, _ = train.feed_data(gt, "sr_bd", 4)
    losses = [tr.optimize_parameters(real_h, ref_l)["loss"] for _ in range(12)]
    assert all(v == v and abs(v) < 1e9 for v in losses)
    assert min(losses[-3:]) < 0.9 * losses[0], losses
    tr.update_learning_rate()
    assert tr.get_current_learning_rate() == 1e-4


def test_captured_step_matches_eager(dev):
    """RescaleTrainer.capture(): the whole optimisation step as one hipGraph.  With the l2 head (no RNG) the replayed steps
    must track eager steps from the same start (not bit-identical: Adam is `capturable` and the gradient planes are summed
    in the same order, but the losses agree far below the step-to-step change)."""
    from selfc_amd import train
    x = load_golden("g8_large_stack")["x"]
    gt = x.reshape(1, T, 3, 32, 48).transpose(1, 2).to(dev)
    real_h, ref_l, _ = train.feed_data(gt, "sr_bd", 4)
    eager = train.RescaleTrainer(_net(dev), dict(train.TRAIN_OPT_LARGE))
    le = [eager.optimize_parameters(real_h, ref_l)["loss"] for _ in range(8)]
    graphed = train.RescaleTrainer(_net(dev), dict(train.TRAIN_OPT_LARGE), capturable=True)
    graphed.capture(real_h, ref_l, warmup=3)                       # 3 real steps
    lg = [graphed.optimize_parameters(real_h, ref_l)["loss"] for _ in range(5)]     # steps 4..8 replayed
    assert all(v == v for v in lg)
    for a, b in zip(le[3:], lg):
        assert abs(a - b) < 2e-2 * abs(a), (le, lg)
    graphed.update_learning_rate()
    assert abs(float(graphed.get_current_learning_rate()) - 1e-4) < 1e-10


def test_train_synthetic_script_one_rank(dev):
    """tools/train_synthetic.py (config 3 launcher) end to end as a child process on this GPU."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "train_synthetic.py"), "--steps", "3", "--warmup", "1",
                          "--global-batch", "2", "--size", "64"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["loss"] == line["loss"]


def test_train_synthetic_data_parallel_path_one_rank(dev):
    """The data-parallel code path on ONE GPU: a one-rank RCCL group, RescaleTrainer(data_parallel=True) - rank-0 broadcast,
    flat gradient buffer, the all-reduce between the two captured hipGraphs - fed by host batches through DevicePrefetcher
    (pinned, H2D on a side stream).  Must land on the same parameters as the plain single-GPU captured step on the same data."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [sys.executable, os.path.join(root, "tools", "train_synthetic.py"), "--steps", "3", "--warmup", "1", "--global-batch", "2", "--size", "64"]
    lines = []
    for extra in (["--dist-1"], ["--dist-1", "--eager", "--host-loader"], []):
        out = subprocess.run(base + extra, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        lines.append(json.loads(out.stdout.strip().splitlines()[-1]))
    dp, dp_host, plain = lines
    assert "ONE all-reduce" in dp["gradient_sync"] and "two hipGraphs" in dp["gradient_sync"] and dp["launch"] == "hipGraph replay"
    assert "DevicePrefetcher" in dp_host["data"] and dp_host["launch"] == "eager"
    assert plain["gradient_sync"] == "single GPU"
    assert all(d["loss"] == d["loss"] and d["value"] > 0 for d in lines)
    assert abs(dp["param_sq_sum"] - plain["param_sq_sum"]) < 1e-6 * plain["param_sq_sum"]      # same seeds, same data, same graphed Adam


def test_parameters_without_gradient_are_skipped_like_stock_autograd(dev):
    """with the flat gradient buffer every parameter owns a zeroed .grad view; a parameter the backward does not
    reach must still be skipped by Adam (no weight decay, no moment decay) exactly as with stock autograd accumulation."""
    from selfc_amd import train
    x = load_golden("g8_large_stack")["x"]
    gt = x.reshape(1, T, 3, 32, 48).transpose(1, 2).to(dev)
    real_h, ref_l, _ = train.feed_data(gt, "sr_bd", 4)
    opt = dict(train.TRAIN_OPT_LARGE, weight_decay_G=0.1)
    after = {}
    for flat in (False, True):
        net = _net(dev)
        net.never_used = torch.nn.Parameter(torch.ones(17, device=dev))
        tr = train.RescaleTrainer(net, opt, flat_grads=flat)
        for _ in range(2):
            tr.optimize_parameters(real_h, ref_l)
        assert net.never_used.grad is None
        after[flat] = net.never_used.detach().clone()
        if flat:
            assert len(tr.sink.untouched()) == 1 and float(tr.sink.flat.abs().max()) > 0
    assert torch.equal(after[False], torch.ones(17, device=dev)) and torch.equal(after[True], after[False])


def test_backward_argument_errors(dev):
    """The gradient entry points refuse bad arguments with SELFC_EINVAL -> RuntimeError, launching nothing."""
    from selfc_amd import _lib, runtime as rt
    L = _lib.lib()
    assert L.selfc_subnet_bwd_scratch_bytes(0, 8, 8, 3, 48) == 0
    x = torch.zeros(64, device=dev)
    with pytest.raises(RuntimeError):
        rt.call("selfc_coupling_bwd", 0, x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), 1.0, 63, _lib.stream_ptr())
    with pytest.raises(RuntimeError):
        rt.call("selfc_lrelu_bwd", x.data_ptr(), x.data_ptr(), 62, _lib.stream_ptr())
    with pytest.raises(RuntimeError):
        rt.call("selfc_freq_fwd_bwd", x.data_ptr(), x.data_ptr(), x.data_ptr(), 1, 6, 8, _lib.stream_ptr())     # H % 4 != 0
    # a CPU tensor in training mode fails loudly (no CPU fallback)
    from selfc_amd.modules.Subnet_constructor import D2DTInput
    m = D2DTInput(3, 48)
    with pytest.raises(RuntimeError):
        m(torch.zeros(7, 3, 8, 8, requires_grad=True))


def test_bench_uvg_script_small(dev):
    """tools/bench_uvg.py (config 5 launcher: GOP split with last-frame padding, shard-by-clip) on a small clip."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "bench_uvg.py"), "--clips", "1", "--frames", "10",
                          "--height", "64", "--width", "96"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["gops_per_clip"] == 2 and line["n_gpus"] == 1 and line["frames_per_s"] > 0


def test_feed_data_variants(dev):
    """feed_data: short clips are padded with their last frame (SelfC_model.py:99-108); 'pytorch_bicubic' (area) LR target."""
    from selfc_amd import train
    gt = torch.rand(2, 3, 5, 16, 24).to(dev)                          # 5 < 7 frames
    real_h, ref_l, clip_len = train.feed_data(gt, "pytorch_bicubic", 4)
    assert clip_len == 5 and real_h.shape == (14, 3, 16, 24) and ref_l.shape == (14, 3, 4, 6)
    assert torch.equal(real_h.reshape(2, 7, 3, 16, 24)[:, 6], gt[:, :, 4])
    want = torch.nn.functional.avg_pool2d(real_h.cpu(), 4)
    assert float((ref_l.cpu() - want).abs().max()) < 1e-6


def test_weight_gradients_do_not_depend_on_side_stream_timing(dev):
    """InvStackFn.backward runs every block's weight-gradient phases on a side stream while the main stream goes on to the next
    block's data phases.  Both use per-slot scratch (absmax, gradient planes): a slot may only be overwritten behind the event
    of its last reader (autograd._SLOT_BUSY) - round 3 joined the streams once, after the loop, and nothing ordered block i's
    weight phase against block i-1's data phase.  Expose that ordering: stall the side stream with a 30-ms delay kernel so the
    main stream runs as far ahead as its dependencies allow; the gradients must be the ones of the undisturbed run, bit for bit
    (the kernels are deterministic)."""
    import ctypes as C
    from selfc_amd import _lib, autograd as ag, train
    x = load_golden("g8_large_stack")["x"]
    gt = x.reshape(1, T, 3, 32, 48).transpose(1, 2).to(dev)
    real_h, ref_l, _ = train.feed_data(gt, "sr_bd", 4)

    def grads(delay_us):
        net = _net(dev)
        tr = train.RescaleTrainer(net, dict(train.TRAIN_OPT_LARGE), flat_params=False)
        tr._zero_grad()
        if delay_us:
            side = ag.side_stream(dev)
            assert side is not None
            clk = torch.zeros(2, dtype=torch.int64, device=dev)
            torch.cuda.synchronize()
            _lib.check(_lib.lib().selfc_profile_clock_sample(clk.data_ptr(), delay_us, C.c_void_p(side.cuda_stream)), "clock_sample")
        tr._forward_backward(real_h, ref_l)
        torch.cuda.synchronize()
        assert not ag._SLOT_BUSY, "slot events must not outlive the backward that recorded them"
        return {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}

    # the multi-stream mode (SELFC_BWD_STREAMS=2; one stream is the default since round 6) with the weight-gradient launches on the
    # side stream per subnet (SELFC_BWD_DEFER_WG=0) - the configuration that HAS a side stream to stall
    old = ag._TWO_STREAMS, ag._DEFER_WG
    ag._TWO_STREAMS, ag._DEFER_WG = True, False
    try:
        ref = grads(0)
        late = grads(30000)
    finally:
        ag._TWO_STREAMS, ag._DEFER_WG = old
    assert ref.keys() == late.keys() and len(ref) > 300
    bad = [n for n in ref if not torch.equal(ref[n], late[n])]
    assert not bad, f"{len(bad)} gradients changed with the side stream delayed, e.g. {bad[:4]}"


def test_flat_optimizer_state_round_trips_through_the_reference_layout(dev):
    """Adam on the ONE flat tensor keeps a single state entry; the reference saves / resumes `optimizers` per
    parameter (base_model.py save_training_state / resume_training).  optimizer_state_dict() must have the per-parameter
    layout a stock Adam over the net's parameters accepts, and load_optimizer_state_dict() must bring a fresh trainer to the
    same next step."""
    from selfc_amd import train
    x = load_golden("g8_large_stack")["x"]
    gt = x.reshape(1, T, 3, 32, 48).transpose(1, 2).to(dev)
    real_h, ref_l, _ = train.feed_data(gt, "sr_bd", 4)
    net = _net(dev)
    tr = train.RescaleTrainer(net, dict(train.TRAIN_OPT_LARGE))
    assert tr.flat_optimizer
    tr.optimize_parameters(real_h, ref_l)
    sd = tr.optimizer_state_dict()
    params = [p for p in net.parameters() if p.requires_grad]
    assert len(sd["state"]) == len(params) == len(sd["param_groups"][0]["params"])
    assert all(sd["state"][i]["exp_avg"].shape == p.shape for i, p in enumerate(params))
    # (a) a stock per-tensor Adam - what the reference constructs - loads it
    stock = torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in params], lr=1e-4)
    stock.load_state_dict(sd)
    assert float(stock.state[stock.param_groups[0]["params"][3]]["step"]) == 1.0
    # (b) resume: a fresh trainer on a copy of the weights, fed that state, takes the same second step
    weights = {k: v.detach().clone() for k, v in net.state_dict().items()}
    net2 = _net(dev)
    net2.load_state_dict(weights)
    tr2 = train.RescaleTrainer(net2, dict(train.TRAIN_OPT_LARGE))
    tr2.load_optimizer_state_dict(sd)
    assert tr2.flat_optimizer
    tr.optimize_parameters(real_h, ref_l)
    tr2.optimize_parameters(real_h, ref_l)
    worst = max(float((a - b).abs().max()) for a, b in zip(net.state_dict().values(), net2.state_dict().values()))
    assert worst < 1e-7, worst
    # (c) a per-tensor state whose steps differ between parameters cannot live on one flat tensor: per-tensor fallback
    sd["state"][0]["step"] = sd["state"][0]["step"] + 1
    tr2.load_optimizer_state_dict(sd)
    assert not tr2.flat_optimizer
    # (d) moving the net after the trainer flattened its parameters is refused, not silently ignored
    net.float().to(dev)                                  # same dtype / device: storage unchanged, still attached
    assert tr.sink.params_attached()
    for p in net.parameters():
        p.data = p.data.clone()
    with pytest.raises(RuntimeError, match="no longer alias"):
        tr.optimize_parameters(real_h, ref_l)


def _reference_layout(sd):
    """An optimizer state as torch 1.7 (the reference's pin) writes it: host tensors, integer `step`, float lr, no capturable /
    foreach / fused keys - what `base_model.save_training_state` leaves in a `.state` file."""
    keep = ("lr", "betas", "eps", "weight_decay", "amsgrad", "initial_lr", "params")
    groups = [{k: (float(v) if torch.is_tensor(v) else v) for k, v in g.items() if k in keep} for g in sd["param_groups"]]
    state = {i: {k: (int(float(v)) if k == "step" else v.detach().cpu().clone()) for k, v in st.items()} for i, st in sd["state"].items()}
    return {"state": state, "param_groups": groups}


def test_reference_optimizer_state_loads_into_a_capturable_trainer(dev):
    """a reference `.state` file (float lr, integer step, no capturable flag) loaded into RescaleTrainer(capturable=True)
    must not replace the device-tensor learning rate or the capturable flag (Adam's step.item() would then fail inside the capture),
    and loaded AFTER capture() it must land in the tensors the captured step updates.  Both orders against an EAGER trainer that
    simply kept training - itself capturable=True, i.e. the same Adam arithmetic: the two Adam flavours differ in the last bit
    of an update, and a 1e-8 weight difference moves the next step's f16-kernel gradients by 0.2 % (LeakyReLU kinks), which Adam's
    early steps turn into +-lr per element (tools/experiments/opt_state_debug2.py; two trainers of ONE flavour agree bit for bit)."""
    from selfc_amd import train
    x = load_golden("g8_large_stack")["x"]
    gt = x.reshape(1, T, 3, 32, 48).transpose(1, 2).to(dev)
    real_h, ref_l, _ = train.feed_data(gt, "sr_bd", 4)
    net_a = _net(dev)
    tr_a = train.RescaleTrainer(net_a, dict(train.TRAIN_OPT_LARGE), capturable=True)      # eager all along (never captured)
    tr_a.optimize_parameters(real_h, ref_l)
    tr_a.optimize_parameters(real_h, ref_l)
    weights2 = {k: v.detach().clone() for k, v in net_a.state_dict().items()}
    sd2 = _reference_layout(tr_a.optimizer_state_dict())
    assert isinstance(sd2["state"][0]["step"], int) and isinstance(sd2["param_groups"][0]["lr"], float) and "capturable" not in sd2["param_groups"][0]
    sd2["param_groups"][0]["lr"] = 5e-5                     # a value the fresh trainer does not have: it must be adopted
    tr_a.optimizer_G.param_groups[0]["lr"].fill_(5e-5)
    tr_a.optimize_parameters(real_h, ref_l)
    weights3 = {k: v.detach().clone() for k, v in net_a.state_dict().items()}
    tr_a.optimize_parameters(real_h, ref_l)                 # A is at step 4

    def worst(n1, n2):
        n2 = n2 if isinstance(n2, dict) else n2.state_dict()
        return max(float((a - b).abs().max()) for a, b in zip(n1.state_dict().values(), n2.values()))

    # (a) load, then capture (one warm-up step = step 3), then one replay = step 4
    net_b = _net(dev)
    net_b.load_state_dict(weights2)
    tr_b = train.RescaleTrainer(net_b, dict(train.TRAIN_OPT_LARGE), capturable=True)
    lr_tensor = tr_b.optimizer_G.param_groups[0]["lr"]
    assert torch.is_tensor(lr_tensor)
    tr_b.load_optimizer_state_dict(sd2)
    grp = tr_b.optimizer_G.param_groups[0]
    assert grp["lr"] is lr_tensor and abs(float(lr_tensor) - 5e-5) < 1e-9 and grp["capturable"] is True
    st = tr_b.optimizer_G.state[grp["params"][0]]
    assert st["step"].is_cuda and float(st["step"]) == 2.0
    assert worst(net_b, weights2) == 0.0
    tr_b.capture(real_h, ref_l, warmup=1)
    torch.cuda.synchronize()
    assert worst(net_b, weights3) < 1e-6, ("after the warm-up step of capture()", worst(net_b, weights3), worst(net_b, weights2))
    tr_b.optimize_parameters(real_h, ref_l)
    torch.cuda.synchronize()
    weights4 = {k: v.detach().clone() for k, v in net_a.state_dict().items()}

    def update_err(net, before, ref_before, ref_after):
        """relative L2 distance between the net's UPDATE (its weights minus `before`) and the reference update: a replayed step and
        an eager step are the same kernels but not bit for bit the same sums, and Adam turns the rounding noise of a gradient that
        is zero in exact arithmetic (proj3 biases) into +-0.1 lr - a per-element bound on the weights would be a lottery; a wrong
        learning rate, step count or moment buffer shows here as an error of order one"""
        num = den = 0.0
        for k, v in net.state_dict().items():
            ref = (ref_after[k] - ref_before[k]).double()
            num += float(((v.detach() - before[k]).double() - ref).pow(2).sum())
            den += float(ref.pow(2).sum())
        return (num / den) ** 0.5
    from conftest import record
    e4 = record("reference optimizer state -> capturable trainer: update error of the first replayed step", update_err(net_b, weights3, weights3, weights4))
    assert e4 < E_UPDATE, ("after the first replay", e4)
    # (b) load AFTER capture: same tensors, new contents; two replays bring B to A's step 4 again
    ids_before = {k: v.data_ptr() for k, v in st.items() if torch.is_tensor(v)}
    with torch.no_grad():
        for k, v in net_b.state_dict().items():
            v.copy_(weights2[k])
    tr_b.load_optimizer_state_dict(sd2)
    st = tr_b.optimizer_G.state[tr_b.optimizer_G.param_groups[0]["params"][0]]
    assert {k: v.data_ptr() for k, v in st.items() if torch.is_tensor(v)} == ids_before
    assert tr_b.optimizer_G.param_groups[0]["lr"] is lr_tensor
    tr_b.optimize_parameters(real_h, ref_l)
    tr_b.optimize_parameters(real_h, ref_l)
    torch.cuda.synchronize()
    e24 = record("reference optimizer state -> capturable trainer: update error of two replayed steps after a reload",
                 update_err(net_b, weights2, weights2, weights4))          # two replayed steps from the reloaded state against A's steps 3 and 4
    assert e24 < E_UPDATE, e24
    # (c) a state that does not cover the captured step's tensors is refused, not half-applied
    bad = {"state": {}, "param_groups": sd2["param_groups"]}
    with pytest.raises(RuntimeError, match="capture again"):
        tr_b.load_optimizer_state_dict(bad)
    # (d) ADVICE r5: a state file with OTHER betas / eps / weight decay after capture() - constants of the captured step - is refused
    import copy
    other = copy.deepcopy(sd2)
    other["param_groups"][0]["betas"] = (0.8, 0.99)
    with pytest.raises(RuntimeError, match="constant of the captured step"):
        tr_b.load_optimizer_state_dict(other)
    assert tuple(tr_b.optimizer_G.param_groups[0]["betas"]) == (0.9, 0.999)


#: bar of the two update errors above (an eager, non-capturable Adam against replayed capturable steps of the same kernels): measured
#: 0.0 and 0.0 since replayed and eager steps are bit-identical (round 5); a wrong lr, step count or moment buffer is an error of order one
E_UPDATE = 1e-6


def test_no_graph_of_the_package_contains_a_memset_node(dev):
    """A memset NODE is not ordered with the kernel nodes around it in a one-stream capture on this runtime (DESIGN 4b, round 5:
    the gradient scale's maximum zeroed late, torch's multi-block reductions left without output).  Every capture of the package -
    the pre-bound round trip, the two-stream whole test path with the STP sampler, the module API's cached graph, the training step
    on one and on three streams - is censused node by node (selfc_graph_stats): kernels and nothing else but the copy nodes of
    the captured tensor copies; zero memsets."""
    from selfc_amd import autograd as ag, pipeline as PL, runtime as rt, train
    from selfc_amd.pipeline import FullTestPath, MultiStreamRoundTrip, RescaleRoundTrip
    x = load_golden("g8_large_stack")["x"]
    gt = x.reshape(1, T, 3, 32, 48).transpose(1, 2).to(dev)
    real_h, ref_l, _ = train.feed_data(gt, "sr_bd", 4)
    old_keep, old_two = rt.KEEP_GRAPHS, ag._TWO_STREAMS
    rt.KEEP_GRAPHS = True
    del rt.GRAPH_LOG[:]
    labels = []
    try:
        net = _net(dev, "gmm").eval()
        xx = torch.cat((real_h, real_h)).contiguous()                      # two clips
        with torch.no_grad():
            r1 = RescaleRoundTrip(net, 14, 32, 48, dev)
            r1.capture(xx)
            r1.replay()
            labels.append("RescaleRoundTrip")
            r2 = MultiStreamRoundTrip(net, 14, 32, 48, dev, 2, part_cls=FullTestPath)
            r2.capture(xx)
            r2.replay()
            labels.append("MultiStreamRoundTrip(FullTestPath) x 2 streams")
            n0 = len(rt.GRAPH_LOG)
            for _ in range(3):                                             # the module API: cached graphs from the second call on
                z, _ = net(x=xx, rev=False)
                net(x=z[:, :3].contiguous(), rev=True)
            labels += ["ModuleGraph"] * (len(rt.GRAPH_LOG) - n0)
            assert len(rt.GRAPH_LOG) > n0 or not PL.MODULE_GRAPH
        for two in (True, False):
            ag._TWO_STREAMS = two
            tr = train.RescaleTrainer(_net(dev, "gmm"), dict(train.TRAIN_OPT_LARGE), capturable=True)
            tr.capture(real_h, ref_l, warmup=2)
            tr.optimize_parameters(real_h, ref_l)
            labels.append("RescaleTrainer.capture, " + ("three streams" if two else "one stream"))
            del tr
        torch.cuda.synchronize()
        assert len(rt.GRAPH_LOG) == len(labels) >= 5, (labels, rt.GRAPH_LOG)
        from conftest import record
        for lab, st in zip(labels, rt.GRAPH_LOG):
            record(f"graph nodes: {lab}", st["nodes"])
            assert st["memset"] == 0 and st["kernel"] > 0 and st["kernel"] + st["memcpy"] + st["other"] == st["nodes"], (lab, st)
    finally:
        rt.KEEP_GRAPHS, ag._TWO_STREAMS = old_keep, old_two
        del rt.GRAPH_LOG[:]


def test_folded_gradient_maxima_change_nothing(dev):
    """max|dOut| of every subnet backward is taken where dOut is produced (coupling gradient kernel, the y1-gradient add, F's dx
    epilogue: selfc_coupling_bwd_x / selfc_add_absmax / dx_amax_out) instead of a pass over dOut per call.  Same maxima, same
    power-of-two scales: the gradients of a step must be the ones of the unfolded path (SELFC_BWD_FOLD_AMAX=0) bit for bit -
    with the third stream (H's chain beside G's) and without it."""
    from selfc_amd import autograd as ag, train
    x = load_golden("g8_large_stack")["x"]
    gt = x.reshape(1, T, 3, 32, 48).transpose(1, 2).to(dev)
    real_h, ref_l, _ = train.feed_data(gt, "sr_bd", 4)

    def grads(fold, two_streams):
        old = ag._FOLD_AMAX, ag._TWO_STREAMS
        ag._FOLD_AMAX, ag._TWO_STREAMS = fold, two_streams
        try:
            net = _net(dev)
            tr = train.RescaleTrainer(net, dict(train.TRAIN_OPT_LARGE), flat_params=False)
            tr._zero_grad()
            tr._forward_backward(real_h, ref_l)
            torch.cuda.synchronize()
            return {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
        finally:
            ag._FOLD_AMAX, ag._TWO_STREAMS = old

    for two in (True, False):
        a, b = grads(False, two), grads(True, two)
        assert a.keys() == b.keys() and len(a) > 300
        bad = [n for n in a if not torch.equal(a[n], b[n])]
        assert not bad, f"streams={two}: {len(bad)} gradients differ, e.g. {bad[:4]}"


@pytest.mark.parametrize("defer", ["launches+finishes", "finishes", "nothing"])
def test_paired_gh_backward_equals_the_two_subnet_calls(dev, defer):
    """Round 6: G and H of a coupling block run their backward as ONE call (selfc_gh_bwd_pair: every launch covers both nets, one
    power-of-two gradient scale from the larger of the two maxima, the input gradient one conv over both nets' planes) and the
    weight-gradient work of the whole stack is deferred to the end of its data-gradient chain (FinJobs: the finishes as one launch per 24
    jobs; with "launches+finishes" the weight-gradient kernels themselves as one launch per kind).  Against round 5's two subnet calls on
    two streams (SELFC_BWD_PAIR=0, finishes per subnet): the same f16 operands up to the exact power-of-two scale, the same fp32 sums
    up to the order of two additions in y1's gradient - which the next subnet rounds to f16 again, so a last-bit difference there is
    a 1e-3 difference of single elements further down the 16 block calls.  Measured: 1.6e-4 worst per-tensor relative L2 (floor: 1e-7
    of the whole gradient); a wrong plane, weight or scale anywhere in the paired path is an error of order one."""
    from conftest import record
    from selfc_amd import autograd as ag, train
    x = load_golden("g8_large_stack")["x"]
    gt = x.reshape(1, T, 3, 32, 48).transpose(1, 2).to(dev)
    real_h, ref_l, _ = train.feed_data(gt, "sr_bd", 4)

    def grads(pair, defer_):
        old = ag._PAIR, ag._DEFER_FIN, ag._DEFER_WG
        ag._PAIR, ag._DEFER_FIN, ag._DEFER_WG = pair, defer_ != "nothing", defer_ == "launches+finishes"
        try:
            net = _net(dev)
            tr = train.RescaleTrainer(net, dict(train.TRAIN_OPT_LARGE), flat_params=False)
            tr._zero_grad()
            losses = tr._forward_backward(real_h, ref_l)
            torch.cuda.synchronize()
            return {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}, [float(v) for v in losses[:2]]
        finally:
            ag._PAIR, ag._DEFER_FIN, ag._DEFER_WG = old

    (a, la), (b, lb) = grads(False, "nothing"), grads(True, defer)
    assert set(a) == set(b) and la == lb
    g_all = float(torch.sqrt(sum((v.double() ** 2).sum() for v in a.values())))
    worst = max(float((a[n] - b[n]).norm() / (a[n].norm() + 1e-7 * g_all)) for n in a)
    record(f"paired G/H backward (deferred: {defer}) vs two subnet calls: worst per-tensor relative L2", worst)
    assert worst < 6e-4, worst


@pytest.mark.parametrize("losstype", ["l2", "l1"])
def test_fused_reconstruction_loss_equals_the_torch_expression(dev, losstype):
    """ReconstructionLoss on selfc_recon_loss (two launches: value + gradient) against the reference's torch expression
    (loss.py:5-21: element-wise value, mean over the four axes one after the other) - the value, and the gradients w.r.t. BOTH
    arguments through a channel-slice view (out[:, :3] of a 51-channel tensor: SelfC_model.py:156) and a scaled sum of two losses."""
    from selfc_amd import train
    g = torch.Generator().manual_seed(5)
    full = torch.rand(14, 51, 9, 12, generator=g).to(dev).requires_grad_(True)
    tgt = torch.rand(14, 3, 9, 12, generator=g).to(dev).requires_grad_(True)
    res = {}
    for fused in (False, True):
        old = train._FUSED_LOSS
        train._FUSED_LOSS = fused
        try:
            full.grad = tgt.grad = None
            crit = train.ReconstructionLoss(losstype)
            loss = (crit(full[:, :3], tgt) * 0.7 + crit(tgt, full[:, 3:6])) * 144 * 144 * 3
            loss.backward()
            res[fused] = (loss.detach().clone(), full.grad.clone(), tgt.grad.clone())
        finally:
            train._FUSED_LOSS = old
    (l0, a0, b0), (l1, a1, b1) = res[False], res[True]
    assert abs(float(l1) - float(l0)) < 2e-6 * abs(float(l0)), (float(l0), float(l1))
    assert float((a1 - a0).abs().max()) < 2e-6 * float(a0.abs().max()) and float((b1 - b0).abs().max()) < 2e-6 * float(b0.abs().max())
    assert float(a1[:, 6:].abs().max()) == 0.0


@pytest.mark.parametrize("capturable", [False, True])
def test_fused_clip_and_adam_equals_torch(dev, capturable):
    """selfc_clip_adam (two launches: norm partials, then clip + Adam on the flat buffer in torch's operation order) against
    GradSink.norm + the clip + torch.optim.Adam.step() on the same flat tensor (SelfC_model.py:172-176): three steps from the same
    weights with the same gradients - the norm clip_grad_norm_ returns, the clipped gradient, the moments and the parameters."""
    from selfc_amd import train
    x = load_golden("g8_large_stack")["x"]
    gt = x.reshape(1, T, 3, 32, 48).transpose(1, 2).to(dev)
    real_h, ref_l, _ = train.feed_data(gt, "sr_bd", 4)
    trs = {}
    for fused in (False, True):
        old = train._FUSED_ADAM
        train._FUSED_ADAM = fused
        try:
            tr = trs[fused] = train.RescaleTrainer(_net(dev), dict(train.TRAIN_OPT_LARGE), capturable=capturable)
            assert tr.flat_optimizer
        finally:
            train._FUSED_ADAM = old
    for step in range(3):
        # one backward, its gradient handed to BOTH trainers (their own backward would differ by LeakyReLU kinks once weights differ by 1e-9)
        trs[False]._zero_grad()
        trs[False]._forward_backward(real_h, ref_l)
        g = trs[False].sink.flat.clone()
        out = {}
        for fused in (False, True):
            old = train._FUSED_ADAM
            train._FUSED_ADAM = fused
            try:
                tr = trs[fused]
                tr.sink.flat.copy_(g)
                tr._clip_and_step()
                st = tr.optimizer_G.state[tr.sink.flat_param]
                out[fused] = (float(tr.grad_norm), tr.sink.flat.clone(), st["exp_avg"].clone(), st["exp_avg_sq"].clone(), tr.sink.flat_param.detach().clone(), float(st["step"]))
            finally:
                train._FUSED_ADAM = old
        a, b = out[False], out[True]
        assert a[5] == b[5] == step + 1
        assert abs(a[0] - b[0]) < 2e-6 * a[0], (a[0], b[0])
        for i, what in ((1, "clipped gradient"), (2, "exp_avg"), (3, "exp_avg_sq")):
            assert float((a[i] - b[i]).abs().max()) <= 4e-6 * float(a[i].abs().max()), (step, what)
        # parameters: the same update up to ONE rounding of p + delta (an ulp of the largest parameter; |update| <= lr = 1e-4)
        assert float((a[4] - b[4]).abs().max()) <= 1.2e-7 * float(a[4].abs().max()), (step, float((a[4] - b[4]).abs().max()))
        with torch.no_grad():                                      # same weights into the next step
            trs[True].sink.flat_param.copy_(trs[False].sink.flat_param)
            for k in ("exp_avg", "exp_avg_sq"):
                trs[True].optimizer_G.state[trs[True].sink.flat_param][k].copy_(trs[False].optimizer_G.state[trs[False].sink.flat_param][k])
        from selfc_amd import runtime as rt
        rt.invalidate_weights()


def test_backward_is_linear_in_the_output_gradient_at_chain_level(dev):
    """A bar the LeakyReLU kinks cannot loosen: with the forward fixed (same input, same weights -> the same f16 features and
    the same masks), the backward of the WHOLE stack (FrequencyAnalyzer + 8 InvBlockExp, selfc_amd.autograd.InvStackFn) is a
    linear map of the output gradient.  grad(a u + b v) must equal a grad(u) + b grad(v) for the input and for every parameter,
    up to the f16 rounding of the gradient planes (each call scales them by its own max|dOut|): relative L2 below 2e-3, ten
    times tighter than the comparisons with autograd through the fp32 oracle can be (Inv_arch.py:20-31, Subnet_constructor.py:
    98-133, SelfC_GMM_arch_inv.py:62-82 are what is differentiated)."""
    net = _net(dev)
    gen = torch.Generator().manual_seed(41)
    x = torch.rand(T, 3, 64, 80, generator=gen).to(dev)
    u = torch.randn(T, 51, 16, 20, generator=gen).to(dev) * 1e-2
    v = torch.randn(T, 51, 16, 20, generator=gen).to(dev) * 1e-2
    a, b = 0.7, -1.9
    names = [n for n, p in net.named_parameters() if n.startswith("operations.") and p.requires_grad]

    def grads(w):
        net.zero_grad(set_to_none=True)
        xd = x.clone().requires_grad_(True)
        z, _ = net(x=xd, rev=False)
        z.backward(w)
        ps = dict(net.named_parameters())
        return xd.grad.detach().clone(), torch.cat([ps[n].grad.detach().reshape(-1) for n in names])

    gx_u, gp_u = grads(u)
    gx_v, gp_v = grads(v)
    gx_w, gp_w = grads(a * u + b * v)
    from conftest import rel_l2
    ex = rel_l2(gx_w, a * gx_u + b * gx_v)
    ep = rel_l2(gp_w, a * gp_u + b * gp_v)
    print("linearity of the stack's backward: input gradient", ex, "parameter gradients", ep)
    assert ex < 2e-3 and ep < 2e-3
    gx_u2, gp_u2 = grads(u)                       # and it is a function: the same call twice, bit for bit
    assert torch.equal(gx_u, gx_u2) and torch.equal(gp_u, gp_u2)


_CHILD_GRADS = r"""
import sys
import numpy as np
import torch
root, out = sys.argv[1], sys.argv[2]
sys.path.insert(0, root)
sys.path.insert(0, root + "/tests")
import test_gpu_train as tg
from selfc_amd import GlobalVar, _lib, train
_lib.lib()
GlobalVar.set_Temporal_LEN(tg.T)
dev = torch.device("cuda:0")
np.savez(out, **{k: v.cpu().numpy() for k, v in tg._chain_case_grads(dev).items()})
"""


def _chain_case_grads(dev):
    """every gradient of one training forward + backward on a ragged clip (latent 19 x 37: tiles with 7 / 5 valid rows, 5 valid columns)"""
    from selfc_amd import train
    gt = torch.rand(1, 3, T, 76, 148, generator=torch.Generator().manual_seed(77)).to(dev)
    real_h, ref_l, _ = train.feed_data(gt, "sr_bd", 4)
    net = _net(dev)
    tr = train.RescaleTrainer(net, dict(train.TRAIN_OPT_LARGE), flat_params=False)
    tr._zero_grad()
    tr._forward_backward(real_h, ref_l)
    torch.cuda.synchronize()
    return {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}


def test_fused_data_gradient_chain_is_bit_identical_to_the_layer_wise_launches(dev, tmp_path):
    """csrc/dgrad_chain.hip (dpre3, dpre2, dpre1 and dx of a dense block as ONE launch, the chain kept in LDS with halo recompute;
    taken while its workgroups fit the chip in one round) against the four or five launches of the generic plane conv it
    replaces (SELFC_BWD_CHAIN=0, in a child process): same fragments, same stage / tap / k order, same MFMA - every gradient of
    a training step must be the same bit for bit, on a ragged frame (backward of Subnet_constructor.py:27-30,126-129)."""
    import os
    import subprocess
    import sys
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out, out2 = str(tmp_path / "grads.npz"), str(tmp_path / "grads_chain.npz")
    # (round 6: by default a G/H pair's chains also run their own dx layers and the two input gradients are added - a different fp32
    # summation of y1's gradient than the layer-wise path's ONE conv over both nets' planes; SELFC_BWD_CHAIN_DX=0 keeps that conv
    # behind the chain, which is the configuration that must be bit-identical to the layer-wise launches)
    subprocess.run([sys.executable, "-c", _CHILD_GRADS, root, out], check=True, env=dict(os.environ, SELFC_BWD_CHAIN="0", SELFC_BWD_CHAIN_DX="0"), timeout=600)
    subprocess.run([sys.executable, "-c", _CHILD_GRADS, root, out2], check=True, env=dict(os.environ, SELFC_BWD_CHAIN_DX="0"), timeout=600)
    with np.load(out) as ref, np.load(out2) as chain:
        assert set(ref.files) == set(chain.files) and len(ref.files) > 300
        bad = [n for n in ref.files if not np.array_equal(chain[n], ref[n])]
        assert not bad, f"{len(bad)} gradients differ, e.g. {bad[:4]}"
        # the default (chains with their own dx layers + one add): the same gradients up to that summation order
        mine = _chain_case_grads(dev)
        g_all = float(np.sqrt(sum((ref[n].astype(np.float64) ** 2).sum() for n in ref.files)))
        worst = max(float(np.linalg.norm(mine[n].cpu().numpy().astype(np.float64) - ref[n]) / (np.linalg.norm(ref[n].astype(np.float64)) + 1e-7 * g_all)) for n in ref.files)
    from conftest import record
    record("chain pair with its own dx layers vs one conv over both nets' planes: worst per-tensor relative L2", worst)
    assert worst < 6e-4, worst


def test_frame_parallel_temporal_conv_is_bit_identical_to_the_frame_walk(dev, tmp_path):
    """Round 6: on small problems the temporal conv5 kernels (forward G/H conv5 + coupling, and conv5^T of the backward) run
    FRAME-PARALLEL - one wave per output frame, its three input frames loaded by that wave - instead of walking a clip's frames in
    sequence (27 us of dependent steps on one 36x36 training septuplet).  Taps and k-steps accumulate in the walk's order, so the
    forward latent, the losses and every gradient of a training step must equal the walk's (SELFC_T5_FP_MAX=0, child process)
    bit for bit (Subnet_constructor.py:106,130-131 and its backward)."""
    import os
    import subprocess
    import sys
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "grads_walk.npz")
    subprocess.run([sys.executable, "-c", _CHILD_GRADS, root, out], check=True, env=dict(os.environ, SELFC_T5_FP_MAX="0"), timeout=600)
    mine = _chain_case_grads(dev)                                    # this process: frame-parallel at this size
    with np.load(out) as ref:
        assert set(ref.files) == set(mine) and len(mine) > 300
        bad = [n for n in mine if not np.array_equal(mine[n].cpu().numpy(), ref[n])]
    assert not bad, f"{len(bad)} gradients differ, e.g. {bad[:4]}"


def test_host_copies_of_weights_after_a_captured_training_leg(dev):
    """on one day 9 of 45 runs of bench.py got ONE 3456-byte weight (`operations.{1,2}.{G,H}.conv1.weight`)
    wrong in the device -> host copy its oracle was fed, always right after the training legs of the same process
    (profiles/r4/parity_leg_host_copy.txt).  The failing sequence, in one process: build the inference net, run it, run captured
    training legs on OTHER nets (graphs, own streams and trainers die when each leg returns), then - with NO device synchronize in
    between, exactly as bench.py's parity leg does - copy the 3456-byte tensors of the first net to the host 200 times, each copy
    checked against the device tensor (copied back, compared on the device) and against a hash taken on the device.  Zero
    mismatches; a mismatch is described (offsets, contents) in the assertion message."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tools")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import bench
    import bench_train
    from selfc_amd.pipeline import RescaleRoundTrip
    net = bench.build_net(dev)
    x = torch.rand(7, 3, 64, 96, device=dev)
    rtp = RescaleRoundTrip(net, 7, 64, 96, dev)
    with torch.no_grad():
        rtp.run(x)
    small = {k: v for k, v in net.state_dict().items() if k.startswith("operations.") and v.numel() * 4 == 3456}
    assert len(small) == 16                                      # G.conv1 and H.conv1 of the eight blocks
    hashes = None
    bad, copies = [], 0
    for leg, lb in enumerate((2, 1)):
        bench_train.run(batch=lb, size=144, steps=6, warmup=1, fh_loss="gmm", profile=False, graph=True)
        # no torch.cuda.synchronize() here on purpose
        for rep in range(100 // len(small) + 1):
            for k, v in small.items():
                c = v.detach().cpu()
                copies += 1
                if not torch.equal(c.to(dev), v):
                    bad.append(bench.describe_bad_copy(k, c, v, {}))
        if hashes is None:
            hashes = {k: bench.tensor_hash(v) for k, v in small.items()}
        for k, v in small.items():                               # and the device tensors themselves never move
            assert bench.tensor_hash(v) == hashes[k] == bench.tensor_hash(v.detach().cpu()), k
    assert copies >= 200
    assert not bad, f"{len(bad)} of {copies} device -> host copies were wrong: {bad[:3]}"



@pytest.mark.parametrize("two_streams", [True, False])
def test_captured_step_reports_the_eager_steps_scalars(dev, two_streams):
    """Round 5: inside a captured step on ONE stream at 8 septuplets per rank the reported l_back_rec came out EQUAL to l_forw_fit
    (loss 987 where 15,909 was due; the gradients, and so the training, were right): the loss was ONE torch mean over 3.5 M elements,
    i.e. torch's multi-block "global reduce" (partials + a semaphore zeroed by a memset in front of the kernel), and on replay that
    kernel left its output block unwritten (tools/experiments/loss_alias_probe.py; SELFC_LOSS_ONE_MEAN=1 brings it back).  The loss
    is now the reference's chained means and the clip's norm is staged (GradSink.norm): no reduction of the captured step has a
    cross-block stage, and no memset node (below).  Held here at the failing size, with and without the side streams: the scalars a
    captured step reports (both losses, their sum, the gradient norm) and the weights it leaves are the eager step's - same kernels,
    same arithmetic, l2 head (no device RNG)."""
    from selfc_amd import GlobalVar, autograd as ag, train
    from selfc_amd.modules.SelfC_GMM_arch_inv import SelfCInvNet
    old = ag._TWO_STREAMS
    ag._TWO_STREAMS = two_streams
    try:
        def make():
            torch.manual_seed(10)
            net = SelfCInvNet(dict(OPT, fh_loss="l2"), 3, 3, "D2DTNet", [4, 4], 2).to(dev)
            return net, train.RescaleTrainer(net, dict(train.TRAIN_OPT_LARGE), capturable=True)
        gt = torch.rand(8, 3, T, 144, 144, generator=torch.Generator().manual_seed(1234)).to(dev)
        real_h, ref_l, _ = train.feed_data(gt, "sr_bd", 4)
        (net_e, tr_e), (net_g, tr_g) = make(), make()
        logs_e, logs_g, norms_e, norms_g = [], [], [], []
        for _ in range(2):
            tr_e.optimize_parameters(real_h, ref_l)
            tr_g.optimize_parameters(real_h, ref_l)
        tr_g.capture(real_h, ref_l, warmup=0)
        for _ in range(4):
            logs_e.append(dict(tr_e.optimize_parameters(real_h, ref_l)))
            norms_e.append(float(tr_e.grad_norm))
            logs_g.append(dict(tr_g.optimize_parameters(real_h, ref_l)))
            norms_g.append(float(tr_g.grad_norm))
        for le, lg, ne, ng in zip(logs_e, logs_g, norms_e, norms_g):
            assert le["l_back_rec"] > 3 * le["l_forw_fit"] > 0                       # the two losses are nowhere near each other here
            for k in ("l_forw_fit", "l_back_rec", "loss"):
                assert abs(le[k] - lg[k]) <= 1e-6 * abs(le[k]), (k, le, lg)
            assert abs(ne - ng) <= 1e-6 * ne, (ne, ng)
            assert abs(le["loss"] - (le["l_forw_fit"] + le["l_back_rec"]) * 144 * 144 * 3) <= 1e-4 * le["loss"]
        # ... and the replayed step IS the eager step: same weights after six steps.  (Until round 5 it was not, on one stream: the
        # maximum that scales the f16 gradient operands was zeroed by hipMemsetAsync, and that memset NODE was not ordered with the
        # kernels around it in the replayed graph - gradient norm 1.5 % off at the first replay, weights 1e-4 apart and growing,
        # tools/experiments/graph_vs_eager_bits.py; csrc/backward.hip bwd_absmax zeroes with a kernel now.)
        worst = max(float((a - b).abs().max()) for a, b in zip(net_e.state_dict().values(), net_g.state_dict().values()))
        assert worst < 1e-7, worst
        # the staged norm is the norm
        assert abs(float(tr_e.sink.norm()) - float(torch.linalg.vector_norm(tr_e.sink.flat.double()))) <= 1e-5 * float(tr_e.sink.norm())
    finally:
        ag._TWO_STREAMS = old
