"""Experiment: host-side enqueue time of one training step vs its GPU time."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from selfc_amd import GlobalVar, train
from selfc_amd.modules.SelfC_GMM_arch_inv import SelfCInvNet
dev = torch.device("cuda:0")
GlobalVar.set_Temporal_LEN(7)
torch.manual_seed(10)
opt = {"global_module": "nonlocal", "stp_blk_num": 6, "fh_loss": "gmm", "scale": 4, "gmm_k": 5}
net = SelfCInvNet(opt, 3, 3, "D2DTNet", [4, 4], 2).to(dev)
tr = train.RescaleTrainer(net, dict(train.TRAIN_OPT_LARGE))
gt = torch.rand(8, 3, 7, 144, 144).to(dev)
real_h, ref_l, _ = train.feed_data(gt, "sr_bd", 4)
for _ in range(3):
    tr.optimize_parameters(real_h, ref_l)
torch.cuda.synchronize()
import cProfile, pstats
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(5):
    tr.optimize_parameters(real_h, ref_l)
pr.disable()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("enqueue+item ms/step", (t1 - t0) / 5 * 1e3, "drain ms", (t2 - t1) * 1e3)
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
