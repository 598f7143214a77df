"""Diagnostic for the whole-suite segfault (hip::Graph::UpdateStreams at the first replay of the captured training step): N
two-stream inference graphs (each with fresh torch.cuda.Stream objects) are captured, replayed and dropped, then the training
step is captured and replayed.  usage: python graph_then_train.py N [hold]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import torch

from conftest import load_golden
from selfc_amd import GlobalVar, _lib, train
from selfc_amd.modules.SelfC_GMM_arch_inv import SelfCInvNet
from selfc_amd.pipeline import FullTestPath, MultiStreamRoundTrip

N = int(sys.argv[1])
hold = len(sys.argv) > 2
T = 7
_lib.lib()
GlobalVar.set_Temporal_LEN(T)
dev = torch.device("cuda:0")
OPT = {"global_module": "nonlocal", "stp_blk_num": 6, "fh_loss": "l2", "scale": 4, "gmm_k": 5}


def net_():
    net = SelfCInvNet(OPT, 3, 3, "D2DTNet", [4, 4], 2)
    sd = {k: v for k, v in load_golden("g8_large_stack").items() if k.startswith("operations.")}
    sd.update({k: v for k, v in load_golden("g7_stp_l2_full_rev").items() if k.startswith("stp_net.")})
    net.load_state_dict(sd, strict=True)
    return net.to(dev)


x = load_golden("g8_large_stack")["x"]
xx = torch.rand(2 * T, 3, 32, 48, device=dev)
net = net_().eval()
kept = []
with torch.no_grad():
    for i in range(N):
        ms = MultiStreamRoundTrip(net, 2 * T, 32, 48, dev, 2, part_cls=FullTestPath)
        ms.capture(xx)
        ms.replay()
        torch.cuda.synchronize()
        if hold:
            kept.append(ms)
print("inference graphs done", N, flush=True)
gt = x.reshape(1, T, 3, 32, 48).transpose(1, 2).to(dev)
real_h, ref_l, _ = train.feed_data(gt, "sr_bd", 4)
graphed = train.RescaleTrainer(net_(), dict(train.TRAIN_OPT_LARGE), capturable=True)
graphed.capture(real_h, ref_l, warmup=2)
for _ in range(3):
    graphed.optimize_parameters(real_h, ref_l)
torch.cuda.synchronize()
print("N", N, "ok", flush=True)
