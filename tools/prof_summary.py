"""Aggregate rocprofv3 CSV output per kernel name.

  python tools/prof_summary.py <dir> [--pmc]
kernel-trace: count / total / average duration per kernel (ns -> us).
--pmc: mean of every counter per kernel (counter_collection.csv)."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void ", "", name)
    return name[:70]


def main():
    d = sys.argv[1]
    pmc = "--pmc" in sys.argv
    if not pmc:
        files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
        agg = defaultdict(lambda: [0, 0.0])
        for f in files:
            for r in csv.DictReader(open(f)):
                dur = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                a = agg[short(r["Kernel_Name"])]
                a[0] += 1
                a[1] += dur
        tot = sum(v[1] for v in agg.values())
        print(f"{'kernel':70s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'%':>6s}")
        for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            print(f"{k:70s} {n:7d} {t / 1e6:10.3f} {t / n / 1e3:9.2f} {100 * t / tot:6.2f}")
    else:
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
        for f in files:
            for r in csv.DictReader(open(f)):
                a = agg[short(r["Kernel_Name"])][r["Counter_Name"]]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
        for k, cs in sorted(agg.items()):
            print(k)
            for c, (n, v) in sorted(cs.items()):
                print(f"    {c:32s} mean/dispatch {v / n:16.1f}   dispatches {n}")


if __name__ == "__main__":
    main()
