"""Experiment: two hipGraphs (separate workspaces) replayed alternately on two streams, so consecutive steps overlap."""
import sys, time, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from selfc_amd.pipeline import MultiStreamRoundTrip

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
net = bench.build_net(dev)
n_frames = bench.B_PER_GPU * bench.T
x = torch.rand(n_frames, 3, bench.H, bench.W, generator=torch.Generator().manual_seed(1234)).to(dev)
nd = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 4
with torch.no_grad():
    runners = [MultiStreamRoundTrip(net, n_frames, bench.H, bench.W, dev, ns) for _ in range(nd)]
    for r in runners:
        r.capture(x)
    streams = [torch.cuda.Stream(device=dev) for _ in range(nd)]
    def run(steps):
        for i in range(steps):
            with torch.cuda.stream(streams[i % nd]):
                runners[i % nd].replay()
    run(6)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(40)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(f"depth {nd} streams {ns}: {40 * bench.B_PER_GPU / dt:.1f} septuplets/s, {dt / 40 * 1e3:.3f} ms/step")
