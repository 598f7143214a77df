"""Does a memset node inside a captured graph execute on every replay on this stack?  selfc_bwd_scale = hipMemsetAsync(amax, 0) +
an atomic-max kernel: capture it once, replay it over inputs whose maximum SHRINKS - without the memset the atomic max keeps the old
value.  Also torch's own global reduce (one sum over 4 M elements) captured and replayed over changing inputs."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from selfc_amd import _lib, runtime as rt
dev = torch.device("cuda:0")
L = _lib.lib()
x = torch.rand(1 << 22, device=dev) * 8
amax = torch.zeros(1, device=dev)
ones = torch.ones(1 << 22, device=dev)
s_out = None
def body():
    global s_out
    _lib.check(L.selfc_bwd_scale(x.data_ptr(), x.numel(), amax.data_ptr(), _lib.stream_ptr()), "selfc_bwd_scale")
    tmp = torch.empty(1 << 20, device=dev)            # allocations around the reduction, as a real capture has
    s_out = (x * ones).sum()
    del tmp
s = rt.warmup_stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    body()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = rt.new_graph()
with rt.graph_capture(g, dev):
    body()
bad = 0
for i in range(12):
    scale = 8.0 / (i + 1)
    x.copy_(torch.rand(1 << 22, device=dev) * scale)
    want_max, want_sum = float(x.max()), float(x.double().sum())
    g.replay(); torch.cuda.synchronize()
    ok_m = float(amax) == want_max
    ok_s = abs(float(s_out) - want_sum) <= 1e-4 * want_sum
    bad += (not ok_m) + (not ok_s)
    print(i, "amax", float(amax), want_max, ok_m, "| sum", float(s_out), want_sum, ok_s, flush=True)
print("mismatches", bad)
