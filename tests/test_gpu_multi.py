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
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["scaling"] == "weak" and d["value"] > 0


def _train2(*extra):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "train_synthetic.py"), "--gpus", "2", "--steps", "2",
                        "--warmup", "1", "--global-batch", "2", "--size", "64", *extra],
                       env=_clean_env(), capture_output=True, text=True, timeout=900)
    d = _json_line(p)
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["global_batch"] == 2
    assert d["param_spread_over_ranks"] == 0.0       # ranks started from DIFFERENT seeds and saw different data
    return d


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
