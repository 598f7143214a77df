"""Config 5 of BASELINE.json: UVG-shaped 100-frame groups at 1080p through the whole test path (forward stack, Quantization,
STP sample, reverse stack), sharded by clip over the GPUs of one node - no data-path collective.

    python tools/bench_uvg.py --clips 2                  # one GPU
    python tools/bench_uvg.py --gpus 8 --clips 8         # starts its own 8 ranks, shard-by-clip (also runs as a rank
                                                         # under an existing torch.distributed.run)

A clip = 100 synthetic frames 3x1080x1920 -> 15 GOPs of 7 (the last one padded by repeating the final frame,
SelfC_model.py:203-209).  Rank r owns clips r, r+world, ...; each GOP is one pipeline.FullTestPath call (latent
270x480).  Rank 0 prints one JSON line: frames/s and 1080p septuplets/s over all ranks, plus the stack's roofline
fractions on the layer-granular byte model of SURVEY 8d."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch  # noqa: E402


def measure(net, dev, clips, frames, H, W, S, ranks, rank=0, world=1):
    """This rank's share of `clips` clips of `frames` frames HxW through the whole test path, S GOPs per hipGraph replay; returns
    the seconds of the timed region (max over ranks) - also what bench.py's `uvg_1080p` leg calls with a bounded sample."""
    from selfc_amd import harness, launch
    from selfc_amd.pipeline import FullTestPath, MultiStreamRoundTrip
    mine = launch.shard(range(clips), rank, world)
    gops = harness.gop_slices(frames)
    S = max(1, S)
    path = MultiStreamRoundTrip(net, 7 * S, H, W, dev, S, part_cls=FullTestPath) if S > 1 else FullTestPath(net, 7, H, W, dev)
    gen = torch.Generator().manual_seed(launch.rank_seed(99, rank))
    clip = torch.rand(frames, 3, H, W, generator=gen).to(dev)                    # one resident clip, reused per owned clip
    work = [g for _ in mine for g in gops]                                       # this rank's GOPs, S per replay
    with torch.no_grad():
        # S GOPs = one hipGraph replay: their frames are copied into the graph's static input first (0.17 GB each, device
        # to device); the last GOP of a clip is padded by repeating its final frame, so every GOP has 7 frames; a last,
        # incomplete group of GOPs re-runs the first ones in its free slots (not counted)
        xs = torch.empty(7 * S, 3, H, W, device=dev)
        for j in range(S):
            xs[7 * j:7 * j + 7].copy_(clip[gops[0]])
        path.capture(xs)
        def all_my_clips():
            for i in range(0, len(work), S):
                for j in range(S):
                    xs[7 * j:7 * j + 7].copy_(clip[work[i + j] if i + j < len(work) else work[j]])
                path.replay()
        path.replay()                                                            # warm-up
        sec = launch.timed_region(all_my_clips, 1, 0, ranks, torch.cuda.synchronize)
    return sec, len(gops)


STACK_FLOP_PX = 2.0 * 267408 * 16          # InvBlock stack fwd + inv per LR pixel-frame
STP_FLOP_PX = 2331776.0                     # STP-large (D2DT subnets, GlobalAgg, GMM head) per LR pixel-frame


def roofline_fracs(ngop, world, sec, H, W):
    """The leg's time against the MFMA peak and the layer-granular HBM model.  `mfma_frac` counts the STACK's FLOPs only (the
    figure rounds 2-4 reported; the STP's time is in the denominator, its work is not in the numerator - which is why it read
    20 % under the 256x448 figure that counted both: VERDICT r4 weak 7); `mfma_frac_whole_path` counts what the leg really
    executes, stack + STP (9,878 GFLOP per 1080p GOP), and is the one to compare with bench.py's full_test_path at 256x448."""
    npx = 7 * (H // 4) * (W // 4)
    per = ngop / world / sec
    return {"mfma_frac": round(STACK_FLOP_PX * npx * per / 1e12 / 2500.0, 4),
            "mfma_frac_whole_path": round((STACK_FLOP_PX + STP_FLOP_PX) * npx * per / 1e12 / 2500.0, 4),
            "hbm_frac_layer_granular": round(2 * 8 * 1815 * 2 * npx * per / 8.0e12, 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=2)
    ap.add_argument("--frames", type=int, default=100)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--gpus", type=int, default=int(os.environ.get("WORLD_SIZE", "1")))
    ap.add_argument("--streams", type=int, default=2, help="GOPs in flight per GPU: one HIP stream each inside one hipGraph "
                    "(GOPs are independent; the HBM-bound temporal convs of one overlap the MFMA-bound fused convs of another)")
    a = ap.parse_args()
    from selfc_amd import launch
    rc = launch.self_launch(a.gpus, os.path.abspath(__file__), sys.argv[1:])      # before anything touches the GPU
    if rc is not None:
        sys.exit(rc)
    # SELFC_BENCH_SHARE_GPU=1: rehearsal on a one-GPU box - every rank on cuda:0, the protocol's collectives over gloo (as bench.py)
    share_gpu = os.environ.get("SELFC_BENCH_SHARE_GPU") == "1" and a.gpus > 1
    local = 0 if share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    ranks = launch.Ranks(a.gpus, "gloo" if share_gpu else "nccl", dev)
    rank, world = ranks.rank, ranks.world
    from selfc_amd import GlobalVar, _lib
    from selfc_amd.modules.SelfC_GMM_arch_inv import SelfCInvNet
    GlobalVar.set_Temporal_LEN(7)
    torch.manual_seed(10)
    opt = {"global_module": "nonlocal", "stp_blk_num": 6, "fh_loss": "gmm", "scale": 4, "gmm_k": 5}
    net = SelfCInvNet(opt, 3, 3, "D2DTNet", [4, 4], 2).to(dev).eval()
    H, W = a.height, a.width
    if H % 4 or W % 4:
        raise SystemExit("frame size must be a multiple of 4 (the reference tiles / pads outside the network)")
    sec, ngops_clip = measure(net, dev, a.clips, a.frames, H, W, a.streams, ranks, rank, world)
    S = max(1, a.streams)
    nranks = ranks.count()
    if rank == 0:
        ngop = a.clips * ngops_clip
        print(json.dumps({"metric": "1080p test path (fwd, quantise, STP, rev), shard-by-clip", "frames_per_s": round(a.clips * a.frames / sec, 2),
                          "gops_per_s": round(ngop / sec, 2), "unit": "7x3x%dx%d septuplets/s" % (H, W), "n_gpus": world, "clips": a.clips,
                          "frames_per_clip": a.frames, "gops_per_clip": ngops_clip, "seconds": round(sec, 3), "dtype": _lib.OPERAND,
                          "stack_roofline_per_gpu": roofline_fracs(ngop, world, sec, H, W),
                          "streams_per_gpu": S, "sharding": f"{world} rank(s), clips round-robin, no data-path collective" + (" - REHEARSAL: all ranks share ONE GPU, collectives over gloo" if share_gpu else ""), "rccl_ranks": nranks, "data": "synthetic"}))
    ranks.close()


if __name__ == "__main__":
    main()
