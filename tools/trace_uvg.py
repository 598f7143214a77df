"""Config 5 under the profiler: 1080p GOPs (7x3x1080x1920, latent 270x480) through the pre-bound pipelines, EAGER, so that
rocprofv3 sees every launch by name (tools/profile_uvg.sh wraps this script; bench_uvg.py is the graph-replayed measurement).

    python3 tools/trace_uvg.py [--path full|stack] [--streams 1|2] [--reps 3] [--height 1080 --width 1920]

`full` = pipeline.FullTestPath (forward stack, Quantization, STP sample, reverse stack: what `uvg_1080p` times), `stack` =
pipeline.RescaleRoundTrip (the headline's unit of work at this size).  With --streams 2 two GOPs run side by side on two HIP
streams, as the bench leg does.  Prints one JSON line: ms per GOP (wall, HIP-synchronised) - under a PMC pass the dispatches are
serialised and the figure means nothing."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--path", choices=("full", "stack"), default="full")
    ap.add_argument("--streams", type=int, default=1)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--warm", type=int, default=1)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    a = ap.parse_args()
    from selfc_amd import GlobalVar
    from selfc_amd.modules.SelfC_GMM_arch_inv import SelfCInvNet
    from selfc_amd.pipeline import FullTestPath, MultiStreamRoundTrip, RescaleRoundTrip
    dev = torch.device("cuda:0")
    GlobalVar.set_Temporal_LEN(7)
    torch.manual_seed(10)
    opt = {"global_module": "nonlocal", "stp_blk_num": 6, "fh_loss": "gmm", "scale": 4, "gmm_k": 5}
    net = SelfCInvNet(opt, 3, 3, "D2DTNet", [4, 4], 2).to(dev).eval()
    cls = FullTestPath if a.path == "full" else RescaleRoundTrip
    S = max(1, a.streams)
    x = torch.rand(7 * S, 3, a.height, a.width, generator=torch.Generator().manual_seed(99)).to(dev)
    path = cls(net, 7, a.height, a.width, dev) if S == 1 else MultiStreamRoundTrip(net, 7 * S, a.height, a.width, dev, S, part_cls=cls)
    with torch.no_grad():
        for _ in range(a.warm):
            path.run(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            path.run(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.reps
    print(json.dumps({"path": a.path, "streams": S, "size": [a.height, a.width], "gops_per_run": S, "ms_per_run": round(dt * 1e3, 3),
                      "frames_per_s": round(7 * S / dt, 1), "launch": "eager"}), flush=True)


if __name__ == "__main__":
    main()
