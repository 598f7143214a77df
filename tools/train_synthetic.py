"""Config 3 of BASELINE.json: the reference's rescaling training loop (train.py + train_rescaling_selfc_large.yml) on
synthetic Vimeo-shaped data, data parallel over RCCL (the reference: DistributedDataParallel around netG,
SelfC_model.py:41-44; global batch 8 split over the ranks, data/__init__.py:13-14).

    python tools/train_synthetic.py --steps 20                   # one GPU
    python tools/train_synthetic.py --gpus 8 --steps 20          # starts its own 8 ranks (one per GPU, RCCL); also runs
                                                                 # as a rank under an existing torch.distributed.run

Default data-parallel path: the plain net + RescaleTrainer(data_parallel): the HIP weight-gradient kernels accumulate into
ONE flat buffer per rank, ONE all-reduce (SUM, / world) of that 13.5 MB buffer per step, and the step is replayed as
hipGraphs ([zero, forward, backward] - all-reduce - [clip, Adam]).  --ddp: the reference's own wrapper
(DistributedDataParallel hooks, eager) as the fallback / cross-check.  --eager: no graph capture.

Data: every rank draws its own 144x144 septuplet crops ON the device (selfc_amd.data.SyntheticSeptuplets, seeded by rank) -
no host-side generation inside the step; --host-loader feeds the same shapes from a host DataLoader-style iterable through
selfc_amd.data.DevicePrefetcher (H2D on a side stream one step ahead), the path real folders / .npy clips take.
Rank 0 prints one JSON line with the aggregate training septuplets/s and every rank's ms per step."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch  # noqa: E402


class _HostBatches:
    """Host-side stand-in for a DataLoader whose worker processes keep up (decoded crops waiting in host memory): a pool of
    `pool` distinct uniform [0,1) (B,C,T,H,W) batches drawn once from a CPU generator and handed out round-robin as pageable
    host tensors - what is measured is the hand-over (pin + H2D one step ahead), not torch.rand on one host thread."""

    def __init__(self, batch, size, seed, n, pool=4):
        gen = torch.Generator().manual_seed(seed)
        self.pool = [torch.rand((batch, 3, 7, size, size), generator=gen) for _ in range(pool)]
        self.n = n

    def __len__(self):
        return self.n

    def __iter__(self):
        for i in range(self.n):
            yield {"GT": self.pool[i % len(self.pool)]}


class _DecodedCrops(torch.utils.data.Dataset):
    """Stand-in for SeptupletDataset with the decode cost left in: every item is built in the WORKER process the way a
    decoded crop is (uint8 frames -> float RGB in [0,1], (C,T,H,W)), from a per-index generator."""

    def __init__(self, size, n, seed):
        self.size, self.n, self.seed = size, n, seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        import numpy as np
        rng = np.random.default_rng(self.seed * 1000003 + i)
        frames = rng.integers(0, 256, size=(7, self.size, self.size, 3), dtype=np.uint8)
        return {"GT": torch.from_numpy(frames.astype(np.float32) / 255.0).permute(3, 0, 1, 2).contiguous()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--global-batch", type=int, default=8, help="septuplets per step over all ranks (yml: batch_size 8)")
    ap.add_argument("--size", type=int, default=144, help="GT_size of the yml")
    ap.add_argument("--gpus", type=int, default=int(os.environ.get("WORLD_SIZE", "1")))
    ap.add_argument("--ddp", action="store_true", help="fallback: wrap the net in DistributedDataParallel as the reference does (eager, no flat gradient buffer)")
    ap.add_argument("--eager", action="store_true", help="no hipGraph capture of the step")
    ap.add_argument("--host-loader", action="store_true", help="host-generated batches through DevicePrefetcher instead of on-device generation")
    ap.add_argument("--workers", type=int, default=0, help="> 0: the reference's loader factory (data.create_dataloader) with this many worker PROCESSES over a synthetic dataset that keeps the per-item host work, through DevicePrefetcher")
    ap.add_argument("--share-gpu", action="store_true", help="all ranks on cuda:0 with a gloo process group (device tensors staged through the host for the collective): "
                    "the data-parallel step - broadcast, flat gradient sink, all-reduce between the two captured graphs - on real kernels where only ONE GPU is visible")
    ap.add_argument("--emulate-ranks", type=int, default=0, help="single process: draw the batches R ranks would draw (their seeds) and train on the concatenation - "
                    "the reference point for --share-gpu (same global batch, same rank-0 weights)")
    ap.add_argument("--fh-loss", default="gmm", choices=("gmm", "l2"), help="STP head (l2: no sampling noise, for run-to-run comparisons)")
    ap.add_argument("--dist-1", action="store_true", help="with --gpus 1: still initialise a one-rank RCCL group and run the data-parallel code path (tests)")
    a = ap.parse_args()
    from selfc_amd import launch
    rc = launch.self_launch(a.gpus, os.path.abspath(__file__), sys.argv[1:])      # before anything touches the GPU
    if rc is not None:
        sys.exit(rc)
    local = 0 if a.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    ranks = launch.Ranks(a.gpus, "gloo" if a.share_gpu else "nccl", dev)
    rank, world = ranks.rank, ranks.world
    import torch.distributed as dist
    solo_group = False
    if world == 1 and a.dist_1:
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{launch.free_port()}", rank=0, world_size=1, device_id=dev)
        solo_group = True
    from selfc_amd import GlobalVar, _lib, data, train
    from selfc_amd.modules.SelfC_GMM_arch_inv import SelfCInvNet
    GlobalVar.set_Temporal_LEN(7)
    torch.manual_seed(10 + rank)      # ranks start DIFFERENT on purpose: the trainer's (or DDP's) rank-0 broadcast must make them equal
    opt = {"global_module": "nonlocal", "stp_blk_num": 6, "fh_loss": a.fh_loss, "scale": 4, "gmm_k": 5}
    net = SelfCInvNet(opt, 3, 3, "D2DTNet", [4, 4], 2).to(dev)
    model, mode = net, "single GPU"
    capture = not a.eager and not a.ddp
    if a.ddp and world > 1:
        from torch.nn.parallel import DistributedDataParallel
        model = DistributedDataParallel(net, device_ids=[local], find_unused_parameters=False)
        mode = "DistributedDataParallel hooks (13.46 MB fp32 in buckets), eager"
    tr = train.RescaleTrainer(model, dict(train.TRAIN_OPT_LARGE), capturable=capture,
                              data_parallel=True if (solo_group and not a.ddp) else None)
    if tr.data_parallel:
        mode = f"flat gradient buffer, ONE all-reduce of {tr.sink.flat.numel() * 4 / 1e6:.2f} MB per step" + \
               (", step = two hipGraphs around it" if capture else ", eager")
    emu = max(1, a.emulate_ranks) if world == 1 else 1
    local_batch = max(1, a.global_batch // (world * emu))
    seed = launch.rank_seed(1234, rank)
    total = a.steps + a.warmup + 4
    if a.workers > 0:
        ds = _DecodedCrops(a.size, total * local_batch * world, seed)
        loader = data.create_dataloader(ds, {"phase": "train", "n_workers": a.workers, "batch_size": local_batch * world},
                                        {"dist": dist.is_initialized(), "gpu_ids": [0]})
        feed = iter(data.DevicePrefetcher(loader, dev, depth=2))
    elif a.host_loader:
        feed = iter(data.DevicePrefetcher(_HostBatches(local_batch, a.size, seed, total), dev, depth=2))
    elif emu > 1:
        srcs = [iter(data.SyntheticSeptuplets(local_batch, 7, a.size, dev, launch.rank_seed(1234, r))) for r in range(emu)]
        feed = iter(lambda: {"GT": torch.cat([next(s_)["GT"] for s_ in srcs], 0)}, None)
    else:
        feed = iter(data.SyntheticSeptuplets(local_batch, 7, a.size, dev, seed))
    log = {}

    def step():
        gt = next(feed)["GT"]                                   # data['GT'] (B,C,T,H,W), already on the device
        real_h, ref_l, _ = train.feed_data(gt, "sr_bd", 4)
        log.update(tr.optimize_parameters(real_h, ref_l))
        tr.update_learning_rate()

    if capture:
        gt = next(feed)["GT"]
        real_h, ref_l, _ = train.feed_data(gt, "sr_bd", 4)
        tr.capture(real_h, ref_l, warmup=2)
    info = {}
    sec = launch.timed_region(step, a.steps, a.warmup, ranks, torch.cuda.synchronize, info)
    my_ms = info["local_seconds"] / a.steps * 1e3          # this rank's own steps (the aggregate uses the slowest rank's time)
    nranks = ranks.count()
    # data parallelism must leave every rank with the same weights (the averaged gradients feed identical Adam steps)
    probe = float(sum(p.detach().double().sum() for p in torch.nn.Module.parameters(net)))
    spread = ranks.max(probe) + ranks.max(-probe)
    checksum = float(sum((p.detach().double() ** 2).sum() for p in torch.nn.Module.parameters(net)))
    per_rank = [None] * world
    if ranks.dist is not None:
        ranks.dist.all_gather_object(per_rank, round(my_ms, 3))
    else:
        per_rank = [round(my_ms, 3)]
    if rank == 0:
        print(json.dumps({"metric": "training septuplets/s (train_rescaling_selfc_large, synthetic 144x144 crops)",
                          "value": round(local_batch * emu * world * a.steps / sec, 2), "unit": "septuplets/s", "n_gpus": world,
                          "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(sec / a.steps * 1e3, 2), "ms_per_step_per_rank": per_rank,
                          "local_batch": local_batch * emu, "global_batch": local_batch * world * emu, "backend": ranks.backend if world > 1 else None, "dtype": _lib.OPERAND,
                          "data": "synthetic, " + (f"DataLoader with {a.workers} worker processes (per-item uint8 -> float work in the workers) through DevicePrefetcher" if a.workers > 0
                                                else "host batches through DevicePrefetcher" if a.host_loader else "generated on the device"),
                          "launch": "hipGraph replay" if capture else "eager", "loss": log.get("loss"), "rccl_ranks": nranks,
                          "param_spread_over_ranks": spread, "param_sq_sum": checksum, "gradient_sync": mode}))
    if solo_group:
        dist.destroy_process_group()
    ranks.close()


if __name__ == "__main__":
    main()
