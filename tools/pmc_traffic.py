"""profiles/rN/pmc_traffic.json from the PMC and kernel-trace summaries of tools/profile_gpu.sh.

  python tools/pmc_traffic.py gpurun_out/prof_<tag> profiles/r4/pmc_traffic.json <source label> [frames per launch = 14]

Per kernel of the hot path (mean per dispatch):
  hbm_bytes_per_launch   FETCH_SIZE x 2 (gfx950 correction, MI355X_MICROARCH.md HBM section) + WRITE_SIZE, both in KiB
  avg_us                 kernel-trace duration (trace pass, not a PMC pass)
  hbm_GBps_measured      hbm_bytes_per_launch / avg_us
  mfma_busy_pct          SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs): share of the launch's cycles in
                         which a SIMD's matrix pipe is busy, averaged over all SIMDs of the chip
  lds_bank_conflict_pct  SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  mfma_executed_over_algorithmic   SQ_VALU_MFMA_BUSY_CYCLES / 32 (cycles of one 32x32x16 MFMA) / (algorithmic MACs of the launch / 16,384):
                         halo recompute, ring tiles and M-tile padding of the fused kernels
  workgroups, cu_share   SQ_WAVES / 8 waves per workgroup, and that over the 256 CUs (one workgroup per CU: the LDS of a fused
                         workgroup fills it) - the share of the chip a launch can occupy at all
  _meta                  configuration of the passes (streams, eager), the commit and the hash of the kernel sources they were
                         taken on: bench.py only reports `traffic` from this file when the sources still hash to that value
"""
import hashlib
import os
import subprocess
import json
import re
import sys

d, out, label = sys.argv[1], sys.argv[2], sys.argv[3]
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 14          # frames per launch: 2 streams x 2 clips of 7
NPX = frames * 64 * 112
# algorithmic MACs per LR pixel-frame of the fused launches (SURVEY 8d / bench.py)
MAC_PX = {"fused_f<0>": 9 * 32 * (48 + 80), "fused_f<1>": 9 * 32 * (112 + 144), "fused_gh": 2 * 9 * 32 * (3 + 35 + 67 + 99)}
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha():
    """the hash bench.py compares: every kernel source and header of csrc/, names included"""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "selfc_amd", "csrc")
    for f in sorted(n_ for n_ in os.listdir(csrc) if n_.endswith((".hip", ".hpp"))):
        h.update(f.encode())
        h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()[:16]

NAMES = {"fused_gh": r"fused_gh_kernel", "fused_f<0>": r"fused_f(16)?_kernel<0", "fused_f<1>": r"fused_f(16)?_kernel<1",
         "conv3x3": r"^conv3x3_kernel<16, 16, 4, 2, 0, false[,>]", "conv5_GH": r"tconv5_kernel<2, 3, 4, 1, 3[,>]", "conv5_F": r"tconv5_kernel<1, 1, 6, 0, 2[,>]",
         "f_couple": r"f_couple_kernel"}


def read(path, counter):
    vals, cur = {}, None
    try:
        for line in open(path):
            if not line.startswith(" "):
                cur = line.strip()
            else:
                m = re.match(r"\s+(\S+)\s+mean/dispatch\s+([0-9.]+)", line)
                if m and m.group(1) == counter:
                    vals[cur] = float(m.group(2))
    except OSError:
        pass
    return vals


def trace(path):
    vals = {}
    try:
        for line in open(path):
            m = re.match(r"(.{70})\s+(\d+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)", line)
            if m:
                vals[m.group(1).strip()] = float(m.group(4))
    except OSError:
        pass
    return vals


def pick(table, pat):
    v = [val for k, val in table.items() if re.search(pat, re.sub(r"^(selfc::)?", "", k))]
    return v[0] if v else None


fetch, write = read(f"{d}/pmc3_summary.txt", "FETCH_SIZE"), read(f"{d}/pmc4_summary.txt", "WRITE_SIZE")
mfma, grbm = read(f"{d}/pmc1_summary.txt", "SQ_VALU_MFMA_BUSY_CYCLES"), read(f"{d}/pmc1_summary.txt", "GRBM_GUI_ACTIVE")
conf, ldsa = read(f"{d}/pmc2_summary.txt", "SQ_LDS_BANK_CONFLICT"), read(f"{d}/pmc2_summary.txt", "SQ_LDS_IDX_ACTIVE")
waves = read(f"{d}/pmc1_summary.txt", "SQ_WAVES")
dur = trace(f"{d}/kernel_trace_summary.txt")
res = {}
for key, pat in NAMES.items():
    f, w = pick(fetch, pat), pick(write, pat)
    if f is None or w is None:
        continue
    fb, wb = f * 1024 * 2, w * 1024
    e = {"fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb, "source": label}
    us = pick(dur, pat)
    if us:
        e["avg_us"] = us
        e["hbm_GBps_measured"] = round((fb + wb) / us / 1e3, 1)
    m, g = pick(mfma, pat), pick(grbm, pat)
    if m is not None and g:
        e["mfma_busy_pct"] = round(100.0 * m / (1024.0 * g / 8.0), 2)
    c, a = pick(conf, pat), pick(ldsa, pat)
    if c is not None and a:
        e["lds_bank_conflict_pct"] = round(100.0 * c / a, 2)
    if m is not None and key in MAC_PX:
        e["mfma_executed_over_algorithmic"] = round(m / 32.0 / (MAC_PX[key] * NPX / 16384.0), 3)
    wv = pick(waves, pat)
    if wv is not None and key in MAC_PX:
        e["workgroups"] = round(wv / 8.0, 1)
        e["cu_share"] = round(min(1.0, wv / 8.0 / 256.0), 3)
    res[key] = e
if "fused_f<0>" in res and "fused_f<1>" in res:      # bench.py times the two launches as one scope
    a, b = res["fused_f<0>"], res["fused_f<1>"]
    res["fused_f"] = {k: a[k] + b[k] for k in ("fetch_bytes_per_launch", "write_bytes_per_launch", "hbm_bytes_per_launch")}
    res["fused_f"]["source"] = label + " (sum of the two launches)"
    if "avg_us" in a and "avg_us" in b:
        res["fused_f"]["avg_us"] = a["avg_us"] + b["avg_us"]
        res["fused_f"]["hbm_GBps_measured"] = round(res["fused_f"]["hbm_bytes_per_launch"] / res["fused_f"]["avg_us"] / 1e3, 1)
    if "mfma_busy_pct" in a and "mfma_busy_pct" in b and "avg_us" in a and "avg_us" in b:
        res["fused_f"]["mfma_busy_pct"] = round((a["mfma_busy_pct"] * a["avg_us"] + b["mfma_busy_pct"] * b["avg_us"]) / (a["avg_us"] + b["avg_us"]), 2)
try:
    commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
except OSError:
    commit = ""
cfg = ""
try:
    cfg = open(f"{d}/config.txt").read().strip()
except OSError:
    pass
res["_meta"] = {"source": label, "config": cfg, "frames_per_launch": frames, "commit": commit, "csrc_sha16": csrc_sha(),
                "note": "counters are means per dispatch of ONE eager step under rocprofv3 --pmc (dispatches serialised by the profiler); durations from the kernel-trace pass"}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
