"""Is a replayed captured step the SAME arithmetic as the eager step?  Two capturable trainers from one seed (l2 head: no RNG), one stepping
eagerly, one replaying its hipGraph; per step the largest weight difference and the reported scalars.  SELFC_BWD_STREAMS=1 for one stream."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from selfc_amd import GlobalVar, train
from selfc_amd.modules.SelfC_GMM_arch_inv import SelfCInvNet
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0"); GlobalVar.set_Temporal_LEN(7)
opt = {"global_module": "nonlocal", "stp_blk_num": 6, "fh_loss": "l2", "scale": 4, "gmm_k": 5}
def make():
    torch.manual_seed(10)
    net = SelfCInvNet(opt, 3, 3, "D2DTNet", [4, 4], 2).to(dev)
    return net, train.RescaleTrainer(net, dict(train.TRAIN_OPT_LARGE), capturable=True)
gt = torch.rand(batch, 3, 7, 144, 144, generator=torch.Generator().manual_seed(1234)).to(dev)
real_h, ref_l, _ = train.feed_data(gt, "sr_bd", 4)
(ne, te), (ng, tg) = make(), make()
for _ in range(2):
    te.optimize_parameters(real_h, ref_l); tg.optimize_parameters(real_h, ref_l)
tg.capture(real_h, ref_l, warmup=0)
out = []
for i in range(6):
    le = te.optimize_parameters(real_h, ref_l); lg = tg.optimize_parameters(real_h, ref_l)
    torch.cuda.synchronize()
    w = max(float((a - b).abs().max()) for a, b in zip(ne.state_dict().values(), ng.state_dict().values()))
    out.append((i + 3, w, le["loss"] - lg["loss"], float(te.grad_norm) - float(tg.grad_norm)))
print("streams", os.environ.get("SELFC_BWD_STREAMS", "2"), "batch", batch, "lib", os.path.basename(os.environ.get("SELFC_LIB", "default")),
      "| (step, max |dw|, d loss, d grad_norm):", [(s, f"{w:.2e}", f"{dl:.3g}", f"{dn:.3g}") for s, w, dl, dn in out], flush=True)
