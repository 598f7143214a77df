"""Does the loss fall?  N optimisation steps on ONE fixed batch (config 3 shapes), loss every few steps, for the launch configurations
of the trainer: python3 tools/experiments/learning_curve.py [--graph] [--batch B] [--steps N] [--fh-loss gmm|l2]"""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
ap = argparse.ArgumentParser()
ap.add_argument("--graph", action="store_true")
ap.add_argument("--batch", type=int, default=4); ap.add_argument("--steps", type=int, default=40); ap.add_argument("--fh-loss", default="gmm")
a = ap.parse_args()
from selfc_amd import GlobalVar, train
from selfc_amd.modules.SelfC_GMM_arch_inv import SelfCInvNet
dev = torch.device("cuda:0"); GlobalVar.set_Temporal_LEN(7); torch.manual_seed(10)
opt = {"global_module": "nonlocal", "stp_blk_num": 6, "fh_loss": a.fh_loss, "scale": 4, "gmm_k": 5}
net = SelfCInvNet(opt, 3, 3, "D2DTNet", [4, 4], 2).to(dev)
tr = train.RescaleTrainer(net, dict(train.TRAIN_OPT_LARGE), capturable=a.graph)
gt = torch.rand(a.batch, 3, 7, 144, 144, generator=torch.Generator().manual_seed(1234)).to(dev)
real_h, ref_l, _ = train.feed_data(gt, "sr_bd", 4)
w0 = net.operations[3].F.conv2.weight.detach().clone()
out = []
for i in range(a.steps):
    if a.graph and i == 2:
        tr.capture(real_h, ref_l, warmup=0)
    log = tr.optimize_parameters(real_h, ref_l)
    if i in (0, 1, 2, 3, 5, 9, 19, a.steps - 1):
        out.append((i + 1, round(log["loss"], 1), round(log["l_forw_fit"], 5), round(log["l_back_rec"], 5)))
torch.cuda.synchronize()
print("graph" if a.graph else "eager", "SELFC_BWD_STREAMS", os.environ.get("SELFC_BWD_STREAMS", "2"), "batch", a.batch, a.fh_loss,
      "| weight moved", float((net.operations[3].F.conv2.weight - w0).abs().max()), "| (step, loss, l_forw, l_back):", out, flush=True)
