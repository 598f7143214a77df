"""One-process-per-GPU launch, sharding and timing protocol shared by bench.py, tools/train_synthetic.py and
tools/bench_uvg.py (and exercised on CPU / gloo by tests/test_multiproc.py).

The reference starts its multi-GPU runs with ``python -m torch.distributed.launch --nproc_per_node N train.py ...``
(/root/reference README.md:85, codes/train.py:19-27: env:// rendezvous, backend nccl).  Here every entry point is
SELF-LAUNCHING: ``python bench.py --gpus N`` with no WORLD_SIZE in the environment starts the N ranks itself
(``self_launch``), before anything has touched the GPU, and relays rank 0's JSON line and the exit code.  Septuplets /
clips are independent units: ranks shard them with no data-path collective; the only collectives are the
timing barrier, the MAX of the elapsed time and a one-element all-reduce that proves every rank is there.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import time
from typing import Callable, List, Optional, Sequence


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launched() -> bool:
    """True when this process is a rank of an existing torch.distributed launch."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def self_launch(nproc: int, script: str, argv: Sequence[str]) -> Optional[int]:
    """Start `nproc` ranks of `script argv` under torch.distributed.run and wait for them; returns the launcher's exit
    code, or None when nothing had to be launched (nproc == 1, or this process already is a rank).

    Must be called BEFORE the process initialises HIP (no torch.cuda call, no library load): the ranks are CHILD
    processes (never an exec of this one), rendezvous on 127.0.0.1 (the container hostname may not resolve)."""
    if nproc <= 1 or launched():
        return None
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script, *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this stack (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.run(cmd, env=env).returncode


class Ranks:
    """Rank identity + the handful of collectives the entry points use.  backend "nccl" (= RCCL over xGMI) on GPUs,
    "gloo" for the CPU rehearsal (tests, --dry-run)."""

    def __init__(self, expect_world: int, backend: str = "nccl", device=None):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != expect_world:
            raise SystemExit(f"--gpus {expect_world} but WORLD_SIZE={self.world}: run `python <script> --gpus {expect_world}` "
                             f"without WORLD_SIZE set (it launches its own ranks) or under torch.distributed.run --nproc-per-node {expect_world}")
        self.backend, self.device, self.dist = backend, device, None
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=device)
            else:
                dist.init_process_group(backend)
            self.dist = dist

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def _tensor(self, v: float):
        import torch
        return torch.tensor([v], dtype=torch.float64, device=self.device if self.backend == "nccl" else "cpu")

    def max(self, v: float) -> float:
        t = self._tensor(v)
        if self.dist is not None:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def count(self) -> int:
        """all-reduce(SUM) of a one per rank: the number of ranks that really took part."""
        t = self._tensor(1.0)
        if self.dist is not None:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return int(round(float(t.item())))

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
            self.dist = None


def shard(units: Sequence, rank: int, world: int) -> List:
    """Round-robin ownership of independent units (clips, septuplets): rank r owns units r, r + world, ..."""
    return list(units[rank::world])


def rank_seed(base: int, rank: int) -> int:
    """Weak scaling: every rank draws its own synthetic septuplets (disjoint data, same shape)."""
    return base + rank


def timed_region(step: Callable[[], None], steps: int, warmup: int, ranks: Ranks,
                 sync: Callable[[], None] = lambda: None, info: Optional[dict] = None) -> float:
    """The benchmark contract: `warmup` untimed steps, then EXACTLY `steps` steps bracketed by barrier + device sync on
    both sides; returns the MAX over ranks of the elapsed seconds.  `info` (optional) receives this rank's own time up to
    its device sync, before the closing barrier ("local_seconds")."""
    for _ in range(warmup):
        step()
    ranks.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    local = time.perf_counter() - t0
    ranks.barrier()
    if info is not None:
        info["local_seconds"] = local
    return ranks.max(time.perf_counter() - t0)


def whole_job_rate(units_per_rank_per_step: float, world: int, steps: int, seconds: float) -> float:
    """units all ranks processed / the slowest rank's time"""
    return units_per_rank_per_step * world * steps / seconds
