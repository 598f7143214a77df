"""bench_train's exact sequence (2 eager steps, capture() with its 3 warm-up steps, replays) with the loss of EVERY step printed."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from selfc_amd import GlobalVar, train
from selfc_amd.modules.SelfC_GMM_arch_inv import SelfCInvNet
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0"); GlobalVar.set_Temporal_LEN(7); torch.manual_seed(10)
opt = {"global_module": "nonlocal", "stp_blk_num": 6, "fh_loss": "gmm", "scale": 4, "gmm_k": 5}
net = SelfCInvNet(opt, 3, 3, "D2DTNet", [4, 4], 2).to(dev)
tr = train.RescaleTrainer(net, dict(train.TRAIN_OPT_LARGE), capturable=True)
gt = torch.rand(batch, 3, 7, 144, 144, generator=torch.Generator().manual_seed(1234)).to(dev)
real_h, ref_l, _ = train.feed_data(gt, "sr_bd", 4)
out = []
for _ in range(2):
    out.append(round(tr.optimize_parameters(real_h, ref_l)["loss"], 1))
tr.capture(real_h, ref_l, warmup=warm)
for i in range(int(os.environ.get("REPLAYS", "12"))):
    log = tr.optimize_parameters(real_h, ref_l)
    out.append((round(log["loss"], 1), round(log["l_forw_fit"], 5), round(log["l_back_rec"], 5)))
print("SELFC_BWD_STREAMS", os.environ.get("SELFC_BWD_STREAMS", "2"), "batch", batch, "capture warmup", warm, out, "grad_norm", float(tr.grad_norm), flush=True)
