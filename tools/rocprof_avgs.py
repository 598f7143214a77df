"""profiles/rN/rocprof_kernel_avgs.json from a rocprofv3 kernel-trace directory (or a kernel_trace_summary.txt of tools/prof_summary.py):
per kernel the number of calls and the average duration in us, + _meta (source hash, the command) - what bench.py shows as
`roofline.rocprof_avg_us` next to its live HIP-event figure.
    python3 tools/rocprof_avgs.py <trace dir | summary.txt> profiles/r6 "<command the trace was taken with>" """
import csv
import glob
import hashlib
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha16():
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "selfc_amd", "csrc")
    for f in sorted(n_ for n_ in os.listdir(csrc) if n_.endswith((".hip", ".hpp"))):
        h.update(f.encode())
        h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()[:16]


def short(name):
    name = re.sub(r"\(anonymous namespace\)::|void |selfc::", "", name)
    m = re.match(r"_ZN\d*_?GLOBAL__N_1\d+([A-Za-z0-9_]+?)(I|E)", name)
    return re.sub(r"\(.*$", "", name) if not m else m.group(1)


def main():
    src, out, cmd = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "")
    agg = defaultdict(lambda: [0, 0.0])
    if os.path.isdir(src):
        for f in glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                agg[k][0] += 1
                agg[k][1] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
    else:
        for line in open(src):
            m = re.match(r"(.{70})\s+(\d+)\s+([0-9.]+)\s+([0-9.]+)", line)
            if m:
                k = short(m.group(1).strip())
                agg[k][0] += int(m.group(2))
                agg[k][1] += float(m.group(3)) * 1e3
    res = {"_meta": {"csrc_sha16": csrc_sha16(), "command": cmd, "source": src}}
    for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        res[k] = {"calls": n, "avg_us": round(us / n, 2)}
    os.makedirs(out, exist_ok=True)
    json.dump(res, open(os.path.join(out, "rocprof_kernel_avgs.json"), "w"), indent=1)
    for k in list(res)[:12]:
        print(k, res[k])


if __name__ == "__main__":
    main()
