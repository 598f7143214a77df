"""Config 5 of BASELINE.json: UVG-shaped 100-frame groups at 1080p through the whole test path (forward stack, Quantization,
STP sample, reverse stack), sharded by clip over the GPUs of one node - no data-path collective.

    python tools/bench_uvg.py --clips 2                                   # one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29512 \\
        tools/bench_uvg.py --clips 8                                       # shard-by-clip

A clip = 100 synthetic frames 3x1080x1920 -> 15 GOPs of 7 (the last one padded by repeating the final frame,
SelfC_model.py:203-209).  Rank r owns clips r, r+world, ...; each GOP is one pipeline.FullTestPath call (latent
270x480).  Rank 0 prints one JSON line: frames/s and 1080p septuplets/s over all ranks, plus the stack's roofline
fractions on the layer-granular byte model of SURVEY 8d."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=2)
    ap.add_argument("--frames", type=int, default=100)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    a = ap.parse_args()
    rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("WORLD_SIZE", 1), ("LOCAL_RANK", 0)))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    from selfc_amd import GlobalVar, _lib, harness
    from selfc_amd.modules.SelfC_GMM_arch_inv import SelfCInvNet
    from selfc_amd.pipeline import FullTestPath
    GlobalVar.set_Temporal_LEN(7)
    torch.manual_seed(10)
    opt = {"global_module": "nonlocal", "stp_blk_num": 6, "fh_loss": "gmm", "scale": 4, "gmm_k": 5}
    net = SelfCInvNet(opt, 3, 3, "D2DTNet", [4, 4], 2).to(dev).eval()
    H, W = a.height, a.width
    if H % 4 or W % 4:
        raise SystemExit("frame size must be a multiple of 4 (the reference tiles / pads outside the network)")
    mine = list(range(rank, a.clips, world))
    gops = harness.gop_slices(a.frames)
    path = FullTestPath(net, 7, H, W, dev)
    gen = torch.Generator().manual_seed(99 + rank)
    clip = torch.rand(a.frames, 3, H, W, generator=gen).to(dev)                  # one resident clip, reused per owned clip
    with torch.no_grad():
        # one GOP = one hipGraph replay: the GOP's frames are copied into the graph's static input first (0.17 GB, device
        # to device); the last GOP of a clip is padded by repeating its final frame, so every GOP has 7 frames
        xs = torch.empty(7, 3, H, W, device=dev)
        xs.copy_(clip[gops[0]])
        path.capture(xs)
        path.replay()                                                            # warm-up
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in mine:
            for g in gops:
                xs.copy_(clip[g])
                path.replay()
        torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    if rank == 0:
        sec = float(dt.item())
        ngop = a.clips * len(gops)
        npx = 7 * (H // 4) * (W // 4)
        flops = 2.0 * 267408 * npx * 16                                          # InvBlock stack fwd + inv per GOP (STP not counted)
        print(json.dumps({"metric": "1080p test path (fwd, quantise, STP, rev), shard-by-clip", "frames_per_s": round(a.clips * a.frames / sec, 2),
                          "gops_per_s": round(ngop / sec, 2), "unit": "7x3x%dx%d septuplets/s" % (H, W), "n_gpus": world, "clips": a.clips,
                          "frames_per_clip": a.frames, "gops_per_clip": len(gops), "seconds": round(sec, 3), "dtype": _lib.OPERAND,
                          "stack_roofline_per_gpu": {"mfma_frac": round(flops * ngop / world / sec / 1e12 / 2500.0, 4),
                                                     "hbm_frac_layer_granular": round(2 * 8 * 1815 * 2 * npx * ngop / world / sec / 8.0e12, 4)},
                          "sharding": f"{world} rank(s), clips round-robin, no data-path collective", "data": "synthetic"}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
