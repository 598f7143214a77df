"""Diagnostic: capture / replay / destroy a multi-branch hipGraph many times in one process.  On the ROCm runtime bundled with
torch 2.10 a long process that had instantiated enough multi-stream graphs segfaulted in hip::Graph::UpdateStreams at the
first replay of a new one (seen in the whole GPU test suite).  usage: python graph_stream_leak.py N [branches] [keep]"""
import sys

import torch

n = int(sys.argv[1])
branches = int(sys.argv[2]) if len(sys.argv) > 2 else 3
keep = len(sys.argv) > 3 and sys.argv[3] == "keep"
dev = torch.device("cuda:0")
x = torch.zeros(1 << 16, device=dev)
streams = [torch.cuda.Stream() for _ in range(branches)]
held = []
for it in range(n):
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cur = torch.cuda.current_stream()
        for s in streams:
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                x.add_(1.0)
        for s in streams:
            cur.wait_stream(s)
    g.replay()
    torch.cuda.synchronize()
    if keep:
        held.append(g)
    if it % 20 == 0:
        print("iteration", it, "ok", flush=True)
print("done", n, flush=True)
