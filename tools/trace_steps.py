"""Steady-state view of a rocprofv3 kernel trace of a REPLAYED training step (tools/bench_train.py --graph):
  python tools/trace_steps.py <dir> [last_steps=5]
Per step (delimited by the forward FrequencyAnalyzer launch): wall span, number of launches, the union of the kernels' intervals
(time with at least one kernel running), the idle remainder (launch gaps on the critical path), the summed kernel time and the time
with exactly 1 / 2 / >= 3 kernels in flight; then the kernels by summed time."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    last = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            nm = re.sub(r"\(anonymous namespace\)::|void |selfc::", "", r["Kernel_Name"])
            rows.append((float(r["Start_Timestamp"]), float(r["End_Timestamp"]), nm[:60], r.get("Queue_Id", "?")))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "freq_fwd_kernel" in r[2]]
    marks = marks[-(last + 1):]
    nst = len(marks) - 1
    agg = defaultdict(lambda: [0, 0.0])
    tot = defaultdict(float)
    for a, b in zip(marks, marks[1:]):
        sel = rows[a:b]
        span = sel[-1][1] - sel[0][0]
        ev = sorted([(s, 1) for s, e, _, _ in sel] + [(e, -1) for s, e, _, _ in sel])
        depth, t_prev, conc = 0, ev[0][0], defaultdict(float)
        for t, k in ev:
            conc[min(depth, 3)] += t - t_prev
            depth += k
            t_prev = t
        tot["span"] += span
        tot["n"] += len(sel)
        tot["sum"] += sum(e - s for s, e, _, _ in sel)
        for k in range(4):
            tot[f"c{k}"] += conc[k]
        for s, e, nm, _ in sel:
            agg[nm][0] += 1
            agg[nm][1] += e - s
    ms = lambda v: v / nst / 1e6  # noqa: E731
    print(f"{nst} replayed steps: {ms(tot['span']):.3f} ms per step, {tot['n'] / nst:.0f} launches, summed kernel time {ms(tot['sum']):.3f} ms")
    print(f"  no kernel running {ms(tot['c0']):.3f} ms | exactly one {ms(tot['c1']):.3f} | two {ms(tot['c2']):.3f} | three or more {ms(tot['c3']):.3f}")
    for nm, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"    {nm:60s} {n / nst:6.1f} x {t / n / 1e3:7.2f} us = {t / nst / 1e6:6.3f} ms/step")


if __name__ == "__main__":
    main()
