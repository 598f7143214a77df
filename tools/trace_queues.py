"""Per-queue view of a rocprofv3 kernel trace (the multi-stream training step):
  python tools/trace_queues.py <dir> [steps] [skip_steps] [kernels listed per queue, default 8]
For the steady-state steps: wall span per step, and per HIP queue the busy time, the number of launches, the idle time
inside its own active window and its five heaviest kernels - i.e. which stream is the critical path and how full it is."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return re.sub(r"void ", "", name)[:56]


def main():
    d = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    skip = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    top = int(sys.argv[4]) if len(sys.argv) > 4 else 8
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((float(r["Start_Timestamp"]), float(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?")))
    rows.sort()
    # steps are delimited by the optimizer's fused Adam kernel (multi_tensor_apply ... the LAST launch group of a step)
    marks = [i for i, r in enumerate(rows) if "freq_fwd_kernel" in r[2]]
    # one forward FrequencyAnalyzer launch per step
    if len(marks) < skip + 2:
        print("not enough steps in the trace", len(marks))
        return
    lo, hi = marks[skip], marks[-1]
    nst = len(marks) - 1 - skip
    sel = rows[lo:hi]
    span = sel[-1][1] - sel[0][0]
    print(f"{nst} steady-state steps, {span / nst / 1e6:.3f} ms per step (trace clock), {len(sel) / nst:.0f} launches per step")
    byq = defaultdict(list)
    for r in sel:
        byq[r[3]].append(r)
    for q, rs in sorted(byq.items(), key=lambda kv: -sum(e - s for s, e, _, _ in kv[1])):
        busy = sum(e - s for s, e, _, _ in rs)
        gaps = [b[0] - a[1] for a, b in zip(rs, rs[1:]) if b[0] > a[1]]
        small = sum(g for g in gaps if g < 20e3)
        print(f"queue {q}: busy {busy / nst / 1e6:.3f} ms/step, {len(rs) / nst:.0f} launches/step, gaps < 20 us between its own kernels "
              f"{small / nst / 1e6:.3f} ms/step ({len([g for g in gaps if g < 20e3]) / nst:.0f}), longer gaps {sum(g for g in gaps if g >= 20e3) / nst / 1e6:.3f} ms/step")
        agg = defaultdict(lambda: [0, 0.0])
        for s, e, k, _ in rs:
            agg[k][0] += 1
            agg[k][1] += e - s
        for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
            print(f"    {k:56s} {n / nst:6.1f} x {t / n / 1e3:7.2f} us = {t / nst / 1e6:6.3f} ms/step")


if __name__ == "__main__":
    main()
