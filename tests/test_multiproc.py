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


@pytest.mark.parametrize("n", [1, 2])
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
