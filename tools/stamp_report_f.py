"""Diagnostic: per-phase time shares of the pairwise-fused F kernels (build with -DSELFC_STAMPS).
Usage on the GPU box:  make -C selfc_amd/csrc diag DIAG=-DSELFC_STAMPS && SELFC_LIB=$PWD/selfc_amd/libselfc_diag.so SELFC_STAMP_DUMP_F=/tmp/stf python tools/stamp_report_f.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from selfc_amd.pipeline import RescaleRoundTrip

dev = torch.device("cuda:0")
net = bench.build_net(dev)
x = torch.rand(28, 3, 256, 448, device=dev)
rt = RescaleRoundTrip(net, 28, 256, 448, dev)
with torch.no_grad():
    for _ in range(2):
        rt.run(x)          # every launch dumps the previous launch's sums of its own pair
torch.cuda.synchronize()
names = ["tile setup", "merged k32 steps", "epilogue 1 + P", "mid barrier", "FM steps (+ halo stores)", "epilogue 2 + P store", "end barrier", "kernel total"]
names1 = ["tile setup", "merged steps incl. chunk barriers", "epilogue 1 + P", "mid barrier", "FM steps (+ halo stores)", "epilogue 2 + P store", "(in the steps)", "kernel total"]
for pair in (0, 1):
    d = np.loadtxt(os.environ["SELFC_STAMP_DUMP_F"] + f".{pair}")
    d = d[d[:, 7] > 0]
    print(f"pair {pair}: waves with data {len(d)}; wave lifetime mean {d[:, 7].mean():.0f} ticks (s_memtime, 100 MHz => {d[:, 7].mean() / 100:.1f} us)")
    tot = d[:, 7].mean()
    w = np.arange(len(d)) % 8  # rows are (workgroup, wave)
    for i, n in enumerate((names1 if pair else names)[:7]):
        ring = d[w < 5, i].mean()      # fused_f16: the ring blocks sit on waves 0..4
        rest = d[w >= 5, i].mean()
        print(f"  {n:26s} {100 * d[:, i].mean() / tot:5.1f} %   (ring waves {100 * ring / tot:5.1f} %, others {100 * rest / tot:5.1f} %)")
    print("  unaccounted: %.1f %%" % (100 * (1 - d[:, :7].sum(1).mean() / tot)))
