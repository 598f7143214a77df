"""Diagnostic: per-phase cycle shares of the fused G/H kernel (build with -DSELFC_STAMPS).
Usage on the GPU box:  SELFC_LIB=$PWD/diag/libselfc_stamps.so SELFC_STAMP_DUMP=/tmp/st.txt python tools/stamp_report.py"""
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
    for _ in range(3):
        rt.run(x)          # every fused launch dumps the previous launch's sums; the last dump is a steady-state launch
torch.cuda.synchronize()
d = np.loadtxt(os.environ["SELFC_STAMP_DUMP"])
d = d[d[:, 4] > 0]
names = os.environ.get("SELFC_STAMP_NAMES", "setup+prefetch,MFMA loop,epilogue,commit+barrier").split(",") + ["kernel total"]
print("waves with data:", len(d))
tot = d[:, 4].mean()
for i, n in enumerate(names):
    print(f"{n:16s} mean {d[:, i].mean():10.0f} cycles  ({100 * d[:, i].mean() / tot:5.1f} % of wave lifetime)   min {d[:, i].min():9.0f} max {d[:, i].max():9.0f}")
print("unaccounted: %.1f %%" % (100 * (1 - d[:, :4].sum(1).mean() / tot)))
