"""SELFC_BWD_STREAMS=1, local batch 8, captured step inside tools/bench_train.run: the reported l_back_rec equals l_forw_fit.
Which tensor holds the wrong value?  The loss module stashes what it returns (and a second opinion through another reduction)."""
import os, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tools"))
import torch
from selfc_amd import train
import bench_train
mode = sys.argv[1] if len(sys.argv) > 1 else "stash"
orig = train.ReconstructionLoss.forward
stash = []
def fwd(self, x, target):
    r = orig(self, x, target)
    if mode == "stash":
        stash.append((self.losstype, r, tuple(x.shape)))
    elif mode == "alt":
        v = torch.sqrt((x - target) ** 2 + self.eps) if self.losstype == "l1" else (x - target) ** 2
        stash.append((self.losstype, r, v.sum() / v.numel()))
    return r
if mode != "plain":
    train.ReconstructionLoss.forward = fwd
os.environ["SELFC_BT_TRACE"] = "1"
out = bench_train.run(batch=int(os.environ.get("B", "8")), size=144, steps=2, warmup=2, fh_loss="gmm", profile=False, graph=True)
print("reported loss", out["loss"])
for e in stash[-2:]:
    print(e[0], "returned", float(e[1]), "ptr", e[1].data_ptr(), "extra", (float(e[2]) if torch.is_tensor(e[2]) else e[2]))
