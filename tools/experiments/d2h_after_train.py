"""Are device -> host copies of a net's weights reliable right after a captured training leg (of ANOTHER net) ended?
bench.py's parity leg once took its oracle weights that way and 9 of 69 runs got ONE 3456-byte tensor wrong in the host copy
(device content intact).  No host copy of the weights is made before the loop (runs that made one early never failed).
  python tools/experiments/d2h_after_train.py [rounds = 8] [legs = train|uvg|full|all]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
import bench_train

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
net = bench.build_net(dev)
from selfc_amd.pipeline import RescaleRoundTrip
x = torch.rand(28, 3, 256, 448, device=dev)
rt = RescaleRoundTrip(net, 28, 256, 448, dev)
with torch.no_grad():
    rt.run(x)
torch.cuda.synchronize()
for r in range(rounds):
    for lb in (8, 1, 2, 4):
        bench_train.run(batch=lb, size=144, steps=20, warmup=2, fh_loss="gmm", profile=False, graph=True)
    bad = []
    copies = {k: v.detach().cpu() for k, v in net.state_dict().items() if k.startswith("operations.")}
    for k, v in net.state_dict().items():
        if k in copies and not torch.equal(copies[k].to(dev), v):
            c2 = v.detach().cpu()
            bad.append((k, tuple(v.shape), "second copy equal to first" if torch.equal(c2, copies[k]) else "second copy differs",
                        int(torch.count_nonzero(c2 - copies[k]))))
    print("round", r, "wrong host copies:", bad, flush=True)
