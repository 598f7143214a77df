"""STP chain alone (STPNet.run_nhwc on the LR frames of config 2), eager on one stream: the workload for
`rocprofv3 --kernel-trace --stats -- python3 tools/trace_stp.py [clips]` (per-kernel cost of the reverse path's STP step).

  clips = septuplets per call: 4 = the whole batch on one stream, 1 = what one stream of the 4-stream pipeline runs.
Prints the wall time per call (device-synchronised) so the kernel sum can be set against it."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from selfc_amd.global_var import GlobalVar                       # noqa: E402
from selfc_amd.modules.SelfC_GMM_arch_inv import STPNet          # noqa: E402

clips = int(sys.argv[1]) if len(sys.argv) > 1 else 4
fh = sys.argv[2] if len(sys.argv) > 2 else "gmm"
iters = int(os.environ.get("ITERS", "10"))
T, h, w = 7, 64, 112
GlobalVar.set_Temporal_LEN(T)
dev = torch.device("cuda:0")
torch.manual_seed(10)
stp = STPNet({"global_module": "nonlocal", "stp_blk_num": 6, "fh_loss": fh, "scale": 4, "gmm_k": 5}).to(dev)
n = clips * T
x1 = torch.rand((n, h * w, 4), device=dev)
hf = torch.empty((n, h * w, 48), device=dev)
eps = torch.empty((n * h * w, 48 * 5), device=dev)
sc = {}
with torch.no_grad():
    for _ in range(3):
        stp.run_nhwc(x1, hf, n, T, h, w, scratch=sc, eps=eps)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        stp.run_nhwc(x1, hf, n, T, h, w, scratch=sc, eps=eps)
    torch.cuda.synchronize()
    print(f"stp chain, {clips} clip(s) of 7x{h}x{w} LR frames, fh_loss {fh}: {(time.perf_counter() - t0) / iters * 1e3:.3f} ms per call ({iters + 3} calls traced)")
