"""The reference's whole test path (SelfCModel.test: forward stack, Quantization, STP sample, reverse stack) as the 4-stream
hipGraph of pipeline.FullTestPath, timed over many replays - the A/B harness for changes to the STP chain.

  python tools/bench_fullpath.py [rounds] [replays]

Run from the root of the tree to be measured (it imports ./selfc_amd); prints one line per round."""
import os
import sys
import time

import torch

sys.path.insert(0, os.getcwd())

from selfc_amd.global_var import GlobalVar                                   # noqa: E402
from selfc_amd.modules.SelfC_GMM_arch_inv import SelfCInvNet                 # noqa: E402
from selfc_amd.pipeline import FullTestPath, MultiStreamRoundTrip            # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
nstreams = int(sys.argv[3]) if len(sys.argv) > 3 else 4
T, H, W, B = 7, 256, 448, 4
GlobalVar.set_Temporal_LEN(T)
dev = torch.device("cuda:0")
torch.manual_seed(10)
opt = {"global_module": "nonlocal", "stp_blk_num": 6, "fh_loss": "gmm", "scale": 4, "gmm_k": 5}
net = SelfCInvNet(opt, 3, 3, "D2DTNet", [4, 4], 2).to(dev).eval()
x = torch.rand((B * T, 3, H, W), generator=torch.Generator().manual_seed(1234)).to(dev)
with torch.no_grad():
    ftp = MultiStreamRoundTrip(net, B * T, H, W, dev, nstreams, part_cls=FullTestPath)
    ftp.capture(x)
    for _ in range(10):
        ftp.replay()
    for r in range(rounds):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            ftp.replay()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        print(f"{os.path.basename(os.getcwd()) or '.'} round {r}: {ms:.3f} ms per batch of {B} = {B / ms * 1e3:.1f} septuplets/s", flush=True)
