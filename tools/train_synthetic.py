"""Config 3 of BASELINE.json: the reference's rescaling training loop (train.py + train_rescaling_selfc_large.yml) on
synthetic Vimeo-shaped data, data parallel over RCCL exactly as the reference does it (DistributedDataParallel around
netG, SelfC_model.py:42; global batch 8 split over the ranks, data/__init__.py:13-14).

    python tools/train_synthetic.py --steps 20                   # one GPU
    python tools/train_synthetic.py --gpus 8 --steps 20          # starts its own 8 ranks (one per GPU, RCCL); also runs
                                                                 # as a rank under an existing torch.distributed.run

Every rank draws its own 144x144 septuplet crops (seeded by rank), runs RescaleTrainer.optimize_parameters (HIP forward,
reverse and backward; DDP all-reduces the 13.46 MB of gradients) and rank 0 prints one JSON line with the aggregate
training septuplets/s."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--global-batch", type=int, default=8, help="septuplets per step over all ranks (yml: batch_size 8)")
    ap.add_argument("--size", type=int, default=144, help="GT_size of the yml")
    ap.add_argument("--gpus", type=int, default=int(os.environ.get("WORLD_SIZE", "1")))
    a = ap.parse_args()
    from selfc_amd import launch
    rc = launch.self_launch(a.gpus, os.path.abspath(__file__), sys.argv[1:])      # before anything touches the GPU
    if rc is not None:
        sys.exit(rc)
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    ranks = launch.Ranks(a.gpus, "nccl", dev)
    rank, world = ranks.rank, ranks.world
    from selfc_amd import GlobalVar, _lib, train
    from selfc_amd.modules.SelfC_GMM_arch_inv import SelfCInvNet
    GlobalVar.set_Temporal_LEN(7)
    torch.manual_seed(10)                                   # same initial weights on every rank (yml manual_seed)
    opt = {"global_module": "nonlocal", "stp_blk_num": 6, "fh_loss": "gmm", "scale": 4, "gmm_k": 5}
    net = SelfCInvNet(opt, 3, 3, "D2DTNet", [4, 4], 2).to(dev)
    model = net
    if world > 1:
        from torch.nn.parallel import DistributedDataParallel
        model = DistributedDataParallel(net, device_ids=[local], find_unused_parameters=False)
    tr = train.RescaleTrainer(model, dict(train.TRAIN_OPT_LARGE))
    local_batch = max(1, a.global_batch // world)
    gen = torch.Generator().manual_seed(launch.rank_seed(1234, rank))
    log = {}

    def step():
        gt = torch.rand(local_batch, 3, 7, a.size, a.size, generator=gen).to(dev, non_blocking=True)     # data['GT'] (B,C,T,H,W)
        real_h, ref_l, _ = train.feed_data(gt, "sr_bd", 4)
        log.update(tr.optimize_parameters(real_h, ref_l))
        tr.update_learning_rate()

    sec = launch.timed_region(step, a.steps, a.warmup, ranks, torch.cuda.synchronize)
    nranks = ranks.count()
    # data parallelism must leave every rank with the same weights (DDP averages the gradients before Adam)
    probe = float(next(net.parameters()).detach().double().sum())
    spread = ranks.max(probe) + ranks.max(-probe)
    if rank == 0:
        print(json.dumps({"metric": "training septuplets/s (train_rescaling_selfc_large, synthetic 144x144 crops)",
                          "value": round(local_batch * world * a.steps / sec, 2), "unit": "septuplets/s", "n_gpus": world,
                          "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(sec / a.steps * 1e3, 2),
                          "local_batch": local_batch, "global_batch": local_batch * world, "dtype": _lib.OPERAND,
                          "data": "synthetic", "loss": log.get("loss"), "rccl_ranks": nranks, "param_spread_over_ranks": spread, "collective": "DDP gradient all-reduce (13.46 MB fp32)" if world > 1 else None}))
    ranks.close()


if __name__ == "__main__":
    main()
