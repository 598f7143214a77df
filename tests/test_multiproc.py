"""N > 1 path of bench.py on CPU: the path shards independent septuplets over ranks with no data-path
collective; the only collectives are the timing barrier and the MAX all-reduce of the elapsed time.
world_size 2 over gloo (runs in the CPU container)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        # weak scaling: every rank owns B_PER_GPU septuplets drawn from its own seed -> disjoint data
        g = torch.Generator().manual_seed(1234 + rank)
        x = torch.rand(2, 3, 8, 8, generator=g)
        steps, dt_local = 5, 0.010 * (rank + 1)          # rank 1 is the slow one
        dist.barrier()
        t = torch.tensor([dt_local], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)          # protocol of bench.py: MAX over ranks
        value = bench.B_PER_GPU * world * steps / float(t.item())
        gathered = [torch.zeros_like(x) for _ in range(world)]
        dist.all_gather(gathered, x)                      # test-only: prove the shards differ
        q.put((rank, float(t.item()), value, bool(torch.equal(gathered[0], gathered[1]))))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharding_and_timing_protocol():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, t0, v0, same0), (r1, t1, v1, same1) = res
    assert abs(t0 - 0.020) < 1e-12 and t0 == t1          # both ranks see the slowest rank's time
    assert v0 == v1 == 4 * 2 * 5 / 0.020                  # whole-job septuplets/s
    assert not same0 and not same1                        # ranks hold different septuplets


def test_bench_refuses_mismatched_world(monkeypatch):
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, WORLD_SIZE="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"],
                       env=env, capture_output=True, text=True)
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)
