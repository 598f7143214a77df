"""aten ops issued by ONE run of the whole test path (pipeline.FullTestPath.run) and of the module API - see count_torch_ops.py."""
import os
import sys
import traceback
from collections import Counter

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

import bench  # noqa: E402
from selfc_amd.pipeline import FullTestPath  # noqa: E402

dev = torch.device("cuda:0")
net = bench.build_net(dev)
x = torch.rand(14, 3, 256, 448, device=dev)
fp = FullTestPath(net, 14, 256, 448, dev)
VIEW = ("view", "reshape", "transpose", "permute", "slice", "select", "expand", "as_strided", "alias", "detach", "unsqueeze", "squeeze",
        "t.default", "_unsafe_view", "unbind", "split", "narrow", "empty", "size", "stride", "is_", "_local_scalar", "lift_fresh", "unfold")


class Count(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.cnt = Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(v in name for v in VIEW):
            fr = [f for f in traceback.extract_stack() if "selfc_amd" in f.filename]
            where = f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno} {fr[-1].line[:70]}" if fr else "(outside the package)"
            self.cnt[(name.replace("aten.", ""), where)] += 1
        return func(*args, **(kwargs or {}))


with torch.no_grad():
    for _ in range(2):
        fp.run(x)
    torch.cuda.synchronize()
    for label, fn in (("FullTestPath.run", lambda: fp.run(x)), ("module API fwd + rev", lambda: net(x=net(x=x, rev=False)[0][:, :3].contiguous(), rev=True))):
        with Count() as c:
            fn()
        torch.cuda.synchronize()
        print(f"# {label}: {sum(c.cnt.values())} device-launching torch ops")
        for (name, where), k in c.cnt.most_common(25):
            print(f"{k:5d}  {name:24s} {where}")
