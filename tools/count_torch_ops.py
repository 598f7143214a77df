"""Which Python lines of the package issue torch (aten) ops during ONE training step of config 3 - the launches that are not
the library's kernels.  TorchDispatchMode + traceback (the profiler's with_stack gives no Python frames on this build); backward
runs in the calling thread (multithreading off) so that the mode sees it."""
import os
import sys
import traceback
from collections import Counter

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

from selfc_amd import GlobalVar, train  # noqa: E402
from selfc_amd.modules.SelfC_GMM_arch_inv import SelfCInvNet  # noqa: E402

dev = torch.device("cuda:0")
GlobalVar.set_Temporal_LEN(7)
torch.manual_seed(10)
opt = {"global_module": "nonlocal", "stp_blk_num": 6, "fh_loss": "gmm", "scale": 4, "gmm_k": 5}
net = SelfCInvNet(opt, 3, 3, "D2DTNet", [4, 4], 2).to(dev)
tr = train.RescaleTrainer(net, dict(train.TRAIN_OPT_LARGE))
gt = torch.rand(8, 3, 7, 144, 144, generator=torch.Generator().manual_seed(1234)).to(dev)
real_h, ref_l, _ = train.feed_data(gt, "sr_bd", 4)
for _ in range(2):
    tr.optimize_parameters(real_h, ref_l)
torch.cuda.synchronize()

VIEW = ("view", "reshape", "transpose", "permute", "slice", "select", "expand", "as_strided", "alias", "detach", "unsqueeze", "squeeze",
        "t.default", "_unsafe_view", "unbind", "split", "narrow", "empty", "size", "stride", "is_", "_local_scalar", "lift_fresh", "unfold")
cnt = Counter()


class Count(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(v in name for v in VIEW):
            fr = [f for f in traceback.extract_stack() if "selfc_amd" in f.filename and "count_torch_ops" not in f.filename]
            where = f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno} {fr[-1].line[:70]}" if fr else "(outside the package)"
            cnt[(name.replace("aten.", ""), where)] += 1
        return func(*args, **(kwargs or {}))


torch.autograd.set_multithreading_enabled(False)
with Count():
    tr.optimize_parameters(real_h, ref_l)
torch.cuda.synchronize()
print("# device-launching torch ops of one training step, by source line:", sum(cnt.values()))
for (name, where), c in cnt.most_common(90):
    print(f"{c:5d}  {name:28s} {where}")
