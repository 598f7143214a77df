"""N > 1 path on CPU (world_size 2 over gloo): the path shards independent septuplets over ranks with no data-path
collective; the only collectives are the timing barrier, the MAX all-reduce of the elapsed time and the rank count.
Every test goes through the code bench.py / tools/* actually run: selfc_amd.launch (Ranks, timed_region, shard,
whole_job_rate, self_launch) and bench.py's own --dry-run leg started the way the driver starts it."""
import json
import os
import subprocess
import sys
import time

import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import bench
    from selfc_amd import launch
    ranks = launch.Ranks(world, "gloo")
    try:
        # weak scaling: every rank owns B_PER_GPU septuplets drawn from its own seed -> disjoint data
        g = torch.Generator().manual_seed(launch.rank_seed(1234, rank))
        x = torch.rand(2, 3, 8, 8, generator=g)
        calls = []

        def step():                       # rank 1 is the slow one
            calls.append(1)
            time.sleep(0.02 * (rank + 1))

        steps, warmup = 4, 2
        dt = launch.timed_region(step, steps, warmup, ranks)
        gathered = [torch.zeros_like(x) for _ in range(world)]
        ranks.dist.all_gather(gathered, x)                      # test-only: prove the shards differ
        q.put((rank, dt, launch.whole_job_rate(bench.B_PER_GPU, world, steps, dt), len(calls), ranks.count(),
               bool(torch.equal(gathered[0], gathered[1])), launch.shard(list(range(7)), rank, world)))
    finally:
        ranks.close()


def test_two_rank_sharding_and_timing_protocol():
    from selfc_amd import launch
    world, port = 2, launch.free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, t0, v0, n0, c0, same0, own0), (_, t1, v1, n1, c1, same1, own1) = res
    assert t0 == t1 and 4 * 0.04 <= t0 < 4 * 0.04 + 0.5      # both ranks report the SLOWEST rank's time for exactly 4 steps
    assert v0 == v1 == 4 * 2 * 4 / t0                          # whole-job septuplets/s: all ranks' units / that time
    assert n0 == n1 == 6                                       # warmup + steps calls, no more
    assert c0 == c1 == 2                                       # all-reduce of ones
    assert not same0 and not same1                             # ranks hold different septuplets
    assert own0 == [0, 2, 4, 6] and own1 == [1, 3, 5]          # round-robin ownership, disjoint and complete


def _run_bench(extra, env=None):
    e = dict(os.environ if env is None else env)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        if env is None:
            e.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *extra], env=e, capture_output=True, text=True, timeout=300)


@pytest.mark.parametrize("n", [1, 2, 8])
def test_bench_self_launches_its_ranks(n):
    """`python bench.py --gpus N` from a clean environment (what the driver runs) must start its own N ranks, print ONE
    JSON line from rank 0 and exit 0.  --dry-run: gloo, no HIP call, so this runs in the CPU container."""
    p = _run_bench(["--gpus", str(n), "--steps", "3", "--warmup", "1", "--dry-run"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["rccl_ranks"] == n and d["dry_run"] is True and d["value"] is None
    assert d["steps"] == 3 and d["steps_counted"] == 3 and d["shards_distinct"] and d["scaling"] == "weak"


def test_bench_relays_a_failing_rank():
    """a rank that dies (here: --gpus 2 ranks told they are a world of 2 but --gpus says 3) must surface as rc != 0"""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="1")
    p = _run_bench(["--gpus", "3", "--steps", "1", "--dry-run"], env=env)
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)


def test_tools_expose_self_launch():
    """tools/train_synthetic.py and tools/bench_uvg.py take --gpus and call launch.self_launch before touching the GPU"""
    for tool in ("train_synthetic.py", "bench_uvg.py"):
        src = open(os.path.join(ROOT, "tools", tool)).read()
        assert "--gpus" in src and "launch.self_launch(" in src
        assert src.index("launch.self_launch(") < src.index("torch.cuda.set_device")


# ---- data-parallel training: ONE all-reduce of the flat gradient buffer (selfc_amd.autograd.GradSink / train.RescaleTrainer) ----

class _ToyNet(torch.nn.Module):
    """CPU stand-in with SelfCInvNet's call signature (x=, rev=) for the HOST logic of RescaleTrainer: `a`'s gradient is
    written straight into the trainer's flat buffer by a hand-rolled backward (as the HIP weight-gradient kernels do, reporting
    None to autograd), `b`'s arrives through autograd, `unused` gets no gradient at all."""

    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(3)
        self.a = torch.nn.Parameter(torch.randn(3, 3, generator=g) * 0.3)
        self.b = torch.nn.Parameter(torch.randn(3, generator=g) * 0.1)
        self.unused = torch.nn.Parameter(torch.ones(5))

    def forward(self, x, rev=False):
        from selfc_amd import autograd as ag
        net = self

        class Mix(torch.autograd.Function):          # y[n,o,h,w] = sum_c a[o,c] x[n,c,h,w]
            @staticmethod
            def forward(ctx, x, a):
                ctx.save_for_backward(x, a)
                return torch.einsum("oc,nchw->nohw", a, x)

            @staticmethod
            def backward(ctx, gy):
                x, a = ctx.saved_tensors
                ga = torch.einsum("nohw,nchw->oc", gy, x)
                sink = ag._SINK
                view = sink.view_of(net.a) if sink is not None else None
                if view is not None:
                    view.add_(ga)                    # "beta = 1" accumulation into the flat buffer
                    ga = None
                return torch.einsum("oc,nohw->nchw", a, gy), ga
        y = Mix.apply(x, self.a) + self.b.view(1, 3, 1, 1)
        if rev:
            return torch.nn.functional.interpolate(y, scale_factor=4.0, mode="nearest"), None
        return torch.nn.functional.avg_pool2d(y, 4), torch.zeros((), dtype=x.dtype)


def _dp_worker(rank, world, port, q, use_ddp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    from selfc_amd import train
    dist.init_process_group("gloo")
    try:
        net = _ToyNet()
        if rank == 1:                                  # ranks start different: the trainer / DDP must broadcast rank 0's weights
            with torch.no_grad():
                net.a.add_(1.0)
        model = torch.nn.parallel.DistributedDataParallel(net, find_unused_parameters=True) if use_ddp else net
        tr = train.RescaleTrainer(model, dict(train.TRAIN_OPT_LARGE), flat_grads=not use_ddp)
        tr.Quantization = lambda v: v                  # the rounding kernel is HIP-only; this is a host-logic test
        assert tr.data_parallel == (not use_ddp) and (tr.sink is None) == use_ddp
        g = torch.Generator().manual_seed(100 + rank)  # different data per rank
        grads = None
        for _ in range(3):
            x = torch.rand(2, 3, 8, 8, generator=g)
            ref = torch.rand(2, 3, 2, 2, generator=g)
            seen = {}
            tr.before_clip = lambda t_: seen.update({n: (None if p.grad is None else p.grad.clone()) for n, p in net.named_parameters()})
            tr.optimize_parameters(x, ref)
            grads = seen
        # plain lists: a tensor in an mp queue is a shared-memory handle that dies with this process
        q.put((rank, {n: p.detach().tolist() for n, p in net.named_parameters()},
               {n: (None if v is None else v.tolist()) for n, v in grads.items()}))
    finally:
        dist.destroy_process_group()


def _run_dp(use_ddp):
    from selfc_amd import launch
    world, port = 2, launch.free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q, use_ddp)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    as_t = lambda d: {n: (None if v is None else torch.tensor(v)) for n, v in d.items()}      # noqa: E731
    return [(r, as_t(ps), as_t(gs)) for r, ps, gs in res]


def test_flat_gradient_all_reduce_equals_ddp_two_ranks():
    """The data-parallel leg of config 3 (SelfC_model.py:41-44) on CPU / gloo, world 2: RescaleTrainer on the plain net averages
    ONE flat gradient buffer per step; the result must be what DistributedDataParallel's hooks give on the same data - same
    averaged gradients, same parameters after three Adam steps, identical on both ranks, and the parameter the backward never
    reaches keeps .grad None (no Adam moment decay / weight decay on it) in both."""
    flat, ddp = _run_dp(False), _run_dp(True)
    (_, p0, g0), (_, p1, g1) = flat
    for n in p0:
        assert torch.equal(p0[n], p1[n]), n                                  # ranks agree bit for bit
    (_, d0, h0), (_, d1, _) = ddp
    for n in d0:
        assert torch.equal(d0[n], d1[n]), n
        assert torch.allclose(p0[n], d0[n], rtol=1e-5, atol=1e-7), n         # flat path == DDP path
    assert g0["unused"] is None and h0["unused"] is None
    assert torch.equal(p0["unused"], torch.ones(5))                          # untouched: not even weight decay
    for n in ("a", "b"):
        assert torch.allclose(g0[n], g1[n]) and torch.allclose(g0[n], h0[n], rtol=1e-5, atol=1e-8), n
    assert float(g0["a"].abs().max()) > 0


def _sink_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    from selfc_amd import autograd as ag
    dist.init_process_group("gloo")
    try:
        ps = [torch.nn.Parameter(torch.zeros(5, 7)), torch.nn.Parameter(torch.zeros(3)), torch.nn.Parameter(torch.zeros(70))]
        sink = ag.GradSink(ps)
        sink.zero()
        sink.view_of(ps[0]).add_(float(rank + 1))                     # a kernel accumulating in place: 1 on rank 0, 2 on rank 1
        (ps[2] * torch.arange(70.0) * (rank + 1)).sum().backward()    # through autograd's AccumulateGrad into the view
        assert ps[2].grad is sink.views[2]
        n_detached = sink.detach_untouched()
        w = sink.all_reduce()
        q.put((rank, w, n_detached, sink.flat.tolist(), ps[1].grad is None, [p.grad is None for p in ps]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_grad_sink_all_reduce_averages_fake_gradients(world):
    """world 8 = the reference's training launch (train.py:19-27, README.md:85: eight ranks, one septuplet each)"""
    from selfc_amd import launch
    port = launch.free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sink_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    mean = (world + 1) / 2.0                                                    # mean of 1..world: 1.5 / 4.5, exact in fp32
    for rank, w, nd, flat, none1, nones in res:
        flat = torch.tensor(flat)
        assert w == world and nd == 1 and none1 and nones == [False, True, False]
        assert flat.numel() == 64 + 64 + 128                                   # 256-byte aligned slices
        assert torch.equal(flat[:35], torch.full((35,), mean))
        assert torch.equal(flat[35:128], torch.zeros(93))                       # pad + the untouched parameter + its pad
        assert torch.allclose(flat[128:198], torch.arange(70.0) * mean)
    assert all(r[3] == res[0][3] for r in res)                                  # every rank holds the same averaged buffer, bit for bit
