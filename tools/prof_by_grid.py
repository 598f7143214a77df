"""Group the launches of one kernel (substring match) in a rocprofv3 kernel trace by grid size: calls / average us.
    python tools/prof_by_grid.py <dir> <kernel substring>"""
import csv
import glob
import os
import sys
from collections import defaultdict

d, pat = sys.argv[1], sys.argv[2]
agg = defaultdict(lambda: [0, 0.0])
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if pat not in r["Kernel_Name"]:
            continue
        key = (r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""), r.get("Workgroup_Size_X", ""))
        a = agg[key]
        a[0] += 1
        a[1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"grid {k}: calls {n:5d}  avg {t / n / 1e3:8.2f} us  total {t / 1e6:8.3f} ms")
