"""Is the CPU oracle (torch fp32 convs) reproducible on this host when torch.set_num_threads changes during the process?
  python tools/experiments/oracle_threads_probe.py fixed|calibrated [reps]
fixed:      set_num_threads(16) once, then `reps` full-size forwards of one septuplet; prints how they differ from the first
calibrated: bench.py's former sequence first (8, 16, 32, 64 threads on a 128x224 crop, then back to 16)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from oracle import selfc_oracle as O

mode = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
T, H, W = 7, 256, 448
net = bench.build_net(torch.device("cpu"))
params = {k: v.detach() for k, v in net.state_dict().items() if k.startswith("operations.")}
x = torch.rand(T, 3, H, W, generator=torch.Generator().manual_seed(1234))
with torch.no_grad():
    if mode == "calibrated":
        small = x[:, :, :128, :224].contiguous()
        for n in (8, 16, 32, 64):
            torch.set_num_threads(n)
            O.large_fwd(params, small, T)
            O.large_fwd(params, small, T)
    torch.set_num_threads(16)
    ref = None
    for i in range(reps):
        t0 = time.perf_counter()
        z = O.large_fwd(params, x, T)
        dt = time.perf_counter() - t0
        if ref is None:
            ref = z
        print(mode, i, "max|z - z0|/max|z0|", float((z - ref).abs().max() / ref.abs().max()), f"{dt:.2f}s", flush=True)
    torch.set_num_threads(1)
    z1 = O.large_fwd(params, x, T)
    print(mode, "one thread vs first", float((z1 - ref).abs().max() / ref.abs().max()), flush=True)
