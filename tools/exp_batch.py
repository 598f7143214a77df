"""Experiment: per-class kernel time per septuplet as a function of the number of septuplets per launch (single stream,
eager) - shows wave-quantisation of the tile grid (784 tiles per 4 septuplets vs 768 resident workgroups)."""
import ctypes as C
import json
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench

from selfc_amd import _lib
from selfc_amd.pipeline import RescaleRoundTrip

dev = torch.device("cuda:0")
net = bench.build_net(dev)
L = _lib.lib()
for B in [int(v) for v in (sys.argv[1:] or ["1", "2", "3", "4", "5", "6", "8"])]:
    n = B * 7
    x = torch.rand(n, 3, 256, 448, device=dev)
    rt = RescaleRoundTrip(net, n, 256, 448, dev)
    with torch.no_grad():
        for _ in range(3):
            rt.run(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            rt.run(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        L.selfc_profile_reset(); L.selfc_profile_enable(1)
        for _ in range(5):
            rt.run(x)
        torch.cuda.synchronize()
        L.selfc_profile_enable(0)
    km = {}
    for cls, name in [(0, "conv3x3"), (1, "conv5_F"), (2, "conv5_GH"), (6, "fused_gh")]:
        ms, cnt = C.c_double(), C.c_longlong()
        L.selfc_profile_read(cls, C.byref(ms), C.byref(cnt))
        km[name] = round(ms.value / 5 / B, 4)
    L.selfc_profile_reset()
    print(json.dumps({"B": B, "ms_per_sept": round(dt * 1e3 / B, 4), "sept_per_s": round(B / dt, 1), "kernel_ms_per_sept": km}))
