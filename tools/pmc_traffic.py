"""profiles/r1/pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE PMC summaries of tools/profile_gpu.sh.

  python tools/pmc_traffic.py gpurun_out/prof_<tag> profiles/r1/pmc_traffic.json <source label>
FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE is doubled (gfx950 correction, MI355X_MICROARCH.md HBM section)."""
import json
import re
import sys

d, out, label = sys.argv[1], sys.argv[2], sys.argv[3]
NAMES = {"fused_gh": r"fused_gh_kernel", "fused_f<0>": r"fused_f_kernel<0>", "fused_f<1>": r"fused_f_kernel<1>",
         "conv3x3": r"^conv3x3_kernel<16, 16, 4, 2, 0, false>", "conv5_GH": r"tconv5_kernel<2, 3, 4, 1, 3>", "conv5_F": r"tconv5_kernel<1, 1, 6, 0, 2>"}


def read(path, counter):
    vals, cur = {}, None
    for line in open(path):
        if not line.startswith(" "):
            cur = line.strip()
        else:
            m = re.match(r"\s+(\S+)\s+mean/dispatch\s+([0-9.]+)", line)
            if m and m.group(1) == counter:
                vals[cur] = float(m.group(2))
    return vals


fetch = read(f"{d}/pmc3_summary.txt", "FETCH_SIZE")
write = read(f"{d}/pmc4_summary.txt", "WRITE_SIZE")
res = {}
for key, pat in NAMES.items():
    f = [v for k, v in fetch.items() if re.search(pat, k)]
    w = [v for k, v in write.items() if re.search(pat, k)]
    if f and w:
        fb, wb = f[0] * 1024 * 2, w[0] * 1024
        res[key] = {"fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb, "source": label}
if "fused_f<0>" in res and "fused_f<1>" in res:      # bench.py times the two launches as one scope
    a, b = res["fused_f<0>"], res["fused_f<1>"]
    res["fused_f"] = {k: a[k] + b[k] for k in ("fetch_bytes_per_launch", "write_bytes_per_launch", "hbm_bytes_per_launch")}
    res["fused_f"]["source"] = label + " (sum of the two launches)"
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
