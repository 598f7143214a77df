"""profiles/rN/uvg1080p_pmc.json from the summaries of tools/profile_uvg.sh (one 1080p GOP, eager, one stream): per hot kernel the
trace duration, HBM bytes (FETCH_SIZE x 2 + WRITE_SIZE, KiB: MI355X_MICROARCH.md HBM section), achieved GB/s, MFMA-busy share and,
where the algorithmic bytes / FLOPs per LR pixel-frame are known, the ratio to them.   python3 tools/pmc_uvg.py profiles/r5"""
import json, os, re, sys

def _csrc_sha16():
    """the hash bench.py compares per counter file: every kernel source and header of csrc/, names included"""
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "selfc_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(n_ for n_ in os.listdir(csrc) if n_.endswith((".hip", ".hpp"))):
        h.update(f.encode())
        h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()[:16]

d = sys.argv[1]
NPX = 7 * 270 * 480
def read(path, counter=None):
    vals, cur = {}, None
    for line in open(path):
        if not line.startswith(" "):
            cur = line.strip()
        else:
            m = re.match(r"\s+(\S+)\s+mean/dispatch\s+([0-9.]+)", line)
            if m and (counter is None or m.group(1) == counter):
                vals.setdefault(cur, {})[m.group(1)] = float(m.group(2))
    return vals
trace = {}
for line in open(os.path.join(d, "uvg1080p_1stream_kernel_trace.txt")):
    m = re.match(r"(.{70})\s+(\d+)\s+([0-9.]+)\s+([0-9.]+)", line)
    if m:
        trace[m.group(1).strip()] = float(m.group(4))
fetch, write, sq = (read(os.path.join(d, f"uvg1080p_pmc_{k}.txt")) for k in ("fetch", "write", "sq"))
NAMES = {"fused_gh": ("selfc::fused_gh_kernel", None, 2 * 9 * 32 * (3 + 35 + 67 + 99)), "fused_f16<0>": ("selfc::fused_f16_kernel<0>", None, 9 * 32 * 128),
         "fused_f16<1>": ("selfc::fused_f16_kernel<1>", None, 9 * 32 * 256), "tconv5_GH": ("tconv5_kernel<2, 3, 4, 1, 3", 1004, None),
         "f_couple": ("selfc::f_couple_kernel", 128, None), "conv3x3 (STP, layer-wise)": ("conv3x3_kernel<16, 16, 4, 2, 0, false", None, None),
         # the largest non-stack launches.  GMM head + sampler: 64->128->256->720 pointwise + sample = 225,280 MAC per
         # pixel; bytes: 64 fp32 features in, 240 fp32 noise values in, 48 fp32 out = 1,408 B.  GlobalAgg mix (consumer = a D2DT block):
         # 64 fp32 channels in, 64 f16 channels out (proj1 64x64 on the MFMA) = 384 B, 4,096 MAC
         "stp_head_gmm (head + sampler)": ("_ZN12_GLOBAL__N_119stp_head_gmm_kernel", 1408, 225280),
         "gagg_mix (GlobalAgg temporal mix + proj1 + residual)": ("_ZN12_GLOBAL__N_115gagg_mix_kernelILb1", 384, 4096),
         "tconv5 (STP conv5)": ("tconv5_kernel<1, 4, 6, 0, 1", None, None)}
out = {"_meta": {"csrc_sha16": _csrc_sha16(), "config": "tools/profile_uvg.sh: ONE 7x3x1080x1920 GOP through pipeline.FullTestPath, eager, one stream; counters = means per dispatch under rocprofv3 --pmc",
                 "px_frames_per_launch": NPX}}
for key, (pat, abytes, mac) in NAMES.items():
    k = next((n for n in trace if n.startswith(pat)), None)
    if k is None:
        continue
    us = trace[k]
    f = next((v for n, v in fetch.items() if n.startswith(pat)), {}).get("FETCH_SIZE")
    w = next((v for n, v in write.items() if n.startswith(pat)), {}).get("WRITE_SIZE")
    s = next((v for n, v in sq.items() if n.startswith(pat)), {})
    e = {"avg_us": us, "ns_per_px_frame": round(us * 1e3 / NPX, 4)}
    if f is not None and w is not None:
        hb = (2 * f + w) * 1024
        e.update({"hbm_bytes_per_launch": hb, "hbm_GBps": round(hb / us / 1e3, 1)})
        if abytes:
            e["hbm_bytes_over_algorithmic"] = round(hb / (abytes * NPX), 3)
    if s.get("GRBM_GUI_ACTIVE"):
        e["mfma_busy_pct"] = round(100 * s["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * s["GRBM_GUI_ACTIVE"] / 8), 1)
    if mac:
        e["algorithmic_TFLOPs"] = round(2 * mac * NPX / us / 1e6, 1)
    out[key] = e
json.dump(out, open(os.path.join(d, "uvg1080p_pmc.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
