"""Concurrency summary of a rocprofv3 kernel trace (the 4-stream hipGraph mode of bench.py):
  python tools/trace_overlap.py <dir> [skip_first_fraction]
Prints, for the steady-state part of the trace: wall span, union of kernel-busy time, sum of kernel durations (their
ratio = average number of kernels in flight), idle gaps, and count / average duration per kernel under concurrency."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return re.sub(r"void ", "", name)[:64]


def main():
    d = sys.argv[1]
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((float(r["Start_Timestamp"]), float(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?")))
    rows.sort()
    selfc = [r for r in rows if not any(k in r[2] for k in ("at::native", "__amd_rocclr", "elementwise_kernel", "rocprim"))]   # the library's kernels
    if not selfc:
        print("no library kernels in the trace")
        return
    t_lo = selfc[0][0] + skip * (selfc[-1][1] - selfc[0][0])     # drop capture / warm-up
    rows = [r for r in selfc if r[0] >= t_lo]
    span = rows[-1][1] - rows[0][0]
    busy, cur_s, cur_e, gaps = 0.0, rows[0][0], rows[0][1], []
    for s, e, _, _ in rows[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append(s - cur_e)
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    total = sum(e - s for s, e, _, _ in rows)
    print(f"kernels {len(rows)}  queues {len(set(r[3] for r in rows))}  span {span / 1e6:.3f} ms  union-busy {busy / 1e6:.3f} ms "
          f"({100 * busy / span:.1f} %)  sum of durations {total / 1e6:.3f} ms  -> {total / busy:.2f} kernels in flight on average")
    if gaps:
        gaps.sort()
        print(f"idle gaps: {len(gaps)}  total {sum(gaps) / 1e3:.1f} us  median {gaps[len(gaps) // 2] / 1e3:.2f} us  max {gaps[-1] / 1e3:.1f} us")
    agg = defaultdict(lambda: [0, 0.0])
    for s, e, k, _ in rows:
        agg[k][0] += 1
        agg[k][1] += e - s
    print(f"{'kernel (durations under concurrency)':64s} {'calls':>6s} {'total_ms':>9s} {'avg_us':>8s} {'% of sum':>8s}")
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:64s} {n:6d} {t / 1e6:9.3f} {t / n / 1e3:8.2f} {100 * t / total:8.2f}")


if __name__ == "__main__":
    main()
