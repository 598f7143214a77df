"""Kernel-trace workload: the reference's whole test path (stack fwd, quantise, STP sample, stack rev) through the module API,
eager, one stream:  rocprofv3 --kernel-trace --stats -d <dir> -- python3 tools/trace_full.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from selfc_amd.modules.Quantization import Quantization

dev = torch.device("cuda:0")
net = bench.build_net(dev)
x = torch.rand(28, 3, 256, 448, device=dev)
q = Quantization()
with torch.no_grad():
    for _ in range(6):
        z, _ = net(x=x, rev=False)
        out = net(x=q(z[:, :3]), rev=True)[0]
torch.cuda.synchronize()
