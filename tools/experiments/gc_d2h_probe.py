"""Does destroying dead hipGraphs (Python's cyclic GC reaching a dead RescaleTrainer) between device -> host copies corrupt
the copies that follow?  bench.py's parity leg once compared against weights copied to the host right after the training leg:
3 of 20 runs got ONE small weight tensor wrong in that copy (device content intact).
  python tools/experiments/gc_d2h_probe.py [collect_after_copy = 10] [rounds = 5]"""
import gc
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
import bench_train

at = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
net = bench.build_net(dev)
w0 = {k: v.detach().cpu().clone() for k, v in net.state_dict().items() if k.startswith("operations.")}
for r in range(rounds):
    gc.disable()
    bench_train.run(batch=1, size=144, steps=5, warmup=2, fh_loss="gmm", profile=False, graph=True)     # dies in reference cycles
    bad, n = [], 0
    for k, v in net.state_dict().items():
        if not k.startswith("operations."):
            continue
        if n == at:
            found = gc.collect()
        c = v.detach().cpu()
        if not torch.equal(c, w0[k]):
            bad.append((n, k, float((c - w0[k]).abs().max())))
        n += 1
    gc.enable()
    print("round", r, "gc.collect() after copy", at, "freed", found, "objects; wrong copies:", bad, flush=True)
