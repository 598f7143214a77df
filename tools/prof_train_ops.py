"""Which torch ops (not the library's kernels) a training step issues: torch.profiler over two steps of config 3,
grouped by op and input shape - to find the small copy / fill / add launches that pad the step."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

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
for _ in range(3):
    tr.optimize_parameters(real_h, ref_l)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    for _ in range(2):
        tr.optimize_parameters(real_h, ref_l)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="count" if False else "self_cuda_time_total", row_limit=45, max_name_column_width=40, max_shapes_column_width=60))
print(prof.key_averages(group_by_stack_n=4).table(sort_by="self_cuda_time_total", row_limit=30, max_name_column_width=30, max_src_column_width=90))

# by source line: which Python lines of this package issue the small torch ops (count per two steps)
from collections import Counter
cnt = Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::add_", "aten::fill_", "aten::sum", "aten::cat", "aten::zero_", "aten::clone", "aten::mul", "aten::add", "aten::sub", "aten::div") and ev.stack:
        line = next((fr for fr in ev.stack if "selfc_amd" in fr or "tools/" in fr), ev.stack[0] if ev.stack else "?")
        cnt[(ev.name, line.strip()[-110:])] += 1
print("\n# small torch ops by source line (two steps)")
for (name, line), c in cnt.most_common(60):
    print(f"{c:5d}  {name:14s} {line}")
