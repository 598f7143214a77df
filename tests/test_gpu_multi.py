"""Multi-GPU legs on real hardware (skipped on the one-GPU test box): the self-launching entry points over RCCL."""
import json
import os
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _need_two():
    if torch.cuda.device_count() < 2:        # device_count() does not initialise HIP in this process
        pytest.skip("needs >= 2 MI355X")


def _clean_env():
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    return e


def _json_line(p):
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_bench_two_gpus_self_launched():
    _need_two()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2"],
                       env=_clean_env(), capture_output=True, text=True, timeout=900)
    d = _json_line(p)
    assert d["n_gpus"] == 2 and d["config"]["rccl_ranks"] == 2 and d["scaling"] == "weak" and d["value"] > 0


def _train2(*extra):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "train_synthetic.py"), "--gpus", "2", "--steps", "2",
                        "--warmup", "1", "--global-batch", "2", "--size", "64", *extra],
                       env=_clean_env(), capture_output=True, text=True, timeout=900)
    d = _json_line(p)
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["global_batch"] == 2
    assert d["param_spread_over_ranks"] == 0.0       # ranks started from DIFFERENT seeds and saw different data
    return d


def test_bench_two_ranks_share_one_gpu():
    """`bench.py --gpus 2` on the ONE visible MI355X (SELFC_BENCH_SHARE_GPU=1: both ranks on cuda:0, the protocol's collectives over
    gloo): self-launch of the child ranks, per-rank synthetic septuplets, capture + replay on real kernels in two processes at
    once, barrier + max-over-ranks timing, one JSON line from rank 0 that says what it is.  Everything of the N-rank bench
    except RCCL; the figure is two ranks SHARING a GPU, so it must not exceed what one rank reaches alone by much."""
    env = dict(_clean_env(), SELFC_BENCH_SHARE_GPU="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "3"],
                       env=env, capture_output=True, text=True, timeout=900)
    d = _json_line(p)
    assert d["n_gpus"] == 2 and d["config"]["rccl_ranks"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert "REHEARSAL" in d["config"]["sharding"]
    assert "full_test_path" not in d and "train_step" not in d and "cpu_baseline" not in d      # the one-rank legs stay off
    assert d["value"] < 2500, d["value"]        # two ranks share one chip: the whole-job rate cannot be twice the one-GPU rate
    print("two ranks on one GPU:", d["value"], "septuplets/s,", d["ms_per_step"], "ms per step (each rank 4 septuplets)")


def test_uvg_shard_by_clip_two_ranks_share_one_gpu():
    """config 5's launcher (`tools/bench_uvg.py --gpus 2`: clips round-robin over the ranks, no data-path collective) with both
    ranks on the ONE visible MI355X: 3 clips of 21 frames at 256x448 -> rank 0 owns clips 0 and 2, rank 1 clip 1; the line counts
    every clip and both ranks."""
    env = dict(_clean_env(), SELFC_BENCH_SHARE_GPU="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_uvg.py"), "--gpus", "2", "--clips", "3", "--frames", "21",
                        "--height", "256", "--width", "448"], env=env, capture_output=True, text=True, timeout=900)
    d = _json_line(p)
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["clips"] == 3 and d["gops_per_clip"] == 3 and d["frames_per_s"] > 0
    assert "REHEARSAL" in d["sharding"]


def test_ddp_two_ranks_keep_equal_parameters():
    """config 3 (train.py under torch.distributed.launch, README.md:85): after optimisation steps on DIFFERENT data the
    ranks' parameters must be identical - the gradient all-reduce over RCCL is the only thing that makes them so.  Three
    legs: the default (flat gradient buffer, one all-reduce, step replayed as two hipGraphs around it), the same eager, and
    the reference's DistributedDataParallel wrapper; the flat path must land where DDP lands."""
    _need_two()
    graphs = _train2()
    assert "ONE all-reduce" in graphs["gradient_sync"] and graphs["launch"] == "hipGraph replay" and len(graphs["ms_per_step_per_rank"]) == 2
    flat = _train2("--eager")
    ddp = _train2("--ddp")
    assert "DistributedDataParallel" in ddp["gradient_sync"]
    assert abs(flat["param_sq_sum"] - ddp["param_sq_sum"]) < 1e-6 * ddp["param_sq_sum"]
    assert abs(graphs["param_sq_sum"] - ddp["param_sq_sum"]) < 1e-5 * ddp["param_sq_sum"]


def _train_line(*extra):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "train_synthetic.py"), "--steps", "2", "--warmup", "1",
                        "--global-batch", "2", "--size", "64", "--fh-loss", "l2", *extra],
                       env=_clean_env(), capture_output=True, text=True, timeout=900)
    return _json_line(p)


def test_two_ranks_on_one_gpu_data_parallel_step():
    """Hardware evidence for the data-parallel step where only ONE MI355X is visible (SelfC_model.py:41-44, data/__init__.py:13-14:
    DistributedDataParallel, global batch split over the ranks): two CHILD ranks share cuda:0 over a gloo process group (device
    tensors staged through the host for the collective - everything except RCCL itself runs as on a multi-GPU node): rank-0
    broadcast, the flat HIP gradient sink, ONE all-reduce of the flat buffer, the step captured as two hipGraphs around it.
    Ranks start from different seeds and draw different data, so identical parameters afterwards can only come from the
    broadcast + the averaged gradients; and the result must be the single-process step on the CONCATENATED batch."""
    if not torch.cuda.device_count():
        pytest.skip("needs an MI355X")
    graphs = _train_line("--gpus", "2", "--share-gpu")
    eager = _train_line("--gpus", "2", "--share-gpu", "--eager")
    one = _train_line("--emulate-ranks", "2")                   # captured: capture()'s two warm-up steps are real steps (5 in all)
    one_eager = _train_line("--emulate-ranks", "2", "--eager")  # 3 steps, like the eager two-rank run
    for d in (graphs, eager):
        assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["backend"] == "gloo" and d["global_batch"] == 2 and d["local_batch"] == 1
        assert "ONE all-reduce" in d["gradient_sync"] and d["param_spread_over_ranks"] == 0.0
        assert d["loss"] == d["loss"]
    assert graphs["launch"] == "hipGraph replay" and "two hipGraphs" in graphs["gradient_sync"] and eager["launch"] == "eager"
    assert one["n_gpus"] == 1 and one["global_batch"] == 2 and one["gradient_sync"] == "single GPU"
    # same rank-0 weights, same global batch, l2 head (no sampling noise): the ranks' averaged gradients are the single process's
    # gradients up to the per-rank gradient scaling (absmax is taken per local batch) and fp32 summation order
    # (bar: one Adam step moves sum(p^2) by about n_params * lr^2 = 0.034 of ~2,085, i.e. 1.6e-5 relative: a missing or doubled step,
    # a gradient that was not averaged or a rank that kept its own weights are all far outside 2e-6)
    print("param_sq_sum: 2 ranks graphs / eager", graphs["param_sq_sum"], eager["param_sq_sum"], "one process graphs / eager", one["param_sq_sum"], one_eager["param_sq_sum"])
    assert abs(graphs["param_sq_sum"] - one["param_sq_sum"]) < 2e-6 * one["param_sq_sum"], (graphs["param_sq_sum"], one["param_sq_sum"])
    assert abs(eager["param_sq_sum"] - one_eager["param_sq_sum"]) < 2e-6 * one_eager["param_sq_sum"], (eager["param_sq_sum"], one_eager["param_sq_sum"])
    assert abs(graphs["loss"] - one["loss"]) < 0.2 * abs(one["loss"])         # the LAST step's loss: rank 0's clip vs the mean over both clips
