"""profiles/rN/train_step_pmc.json from the summaries of tools/pmc_train.sh: for the kernels that make up most of the training step's
kernel time - launches and time per step (kernel trace, eager step), MFMA-busy share, wave-cycle split (issue / wait), LDS
bank-conflict share, HBM bytes (FETCH_SIZE x 2 + WRITE_SIZE, KiB) and GB/s per launch.
    python3 tools/pmc_train.py gpurun_out/pmc_train profiles/r5 [steps traced = 8]"""
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

d, out = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
def read(path):
    vals, cur = {}, None
    for line in open(path):
        if not line.startswith(" "):
            cur = line.strip()
        else:
            m = re.match(r"\s+(\S+)\s+mean/dispatch\s+([0-9.]+)", line)
            if m:
                vals.setdefault(cur, {})[m.group(1)] = float(m.group(2))
    return vals
trace = []
for line in open(os.path.join(d, "kernel_trace_summary.txt")):
    m = re.match(r"(.{70})\s+(\d+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)", line)
    if m:
        trace.append((m.group(1).strip(), int(m.group(2)), float(m.group(3)), float(m.group(4))))
p = [read(os.path.join(d, f"pmc{i}_summary.txt")) for i in (1, 2, 3, 4)]
res = {"_meta": {"csrc_sha16": _csrc_sha16(), "config": "tools/pmc_train.sh: tools/bench_train.py --batch 8 (8 x 7x3x144x144, fh_loss gmm), EAGER step; durations from a kernel-trace "
                           f"pass over {steps} steps, counters = means per dispatch of one step under rocprofv3 --pmc", "steps_traced": steps}}
hot = [t for t in trace if t[0].startswith(("conv3x3_kernel", "wgrad", "tconv5_kernel", "selfc::fused", "selfc::dgrad", "coupling_bwd", "_ZN12_GLOBAL__N_121grad_to_planes"))][:9]
for name, calls, total_ms, avg in hot:
    e = {"launches_per_step": round(calls / steps, 1), "avg_us": avg, "ms_per_step": round(total_ms / steps, 3)}
    g = lambda i, c: next((v.get(c) for n, v in p[i].items() if n.startswith(name[:60])), None)   # noqa: E731
    gui, busy, wc = g(0, "GRBM_GUI_ACTIVE"), g(0, "SQ_VALU_MFMA_BUSY_CYCLES"), g(0, "SQ_WAVE_CYCLES")
    if gui and busy is not None:
        e["mfma_busy_pct"] = round(100 * busy / (1024 * gui / 8), 1)
    if wc:
        e["wave_cycles_issuing_pct"] = round(100 * (g(0, "SQ_ACTIVE_INST_ANY") or 0) / wc, 1)
        e["wave_cycles_waiting_pct"] = round(100 * (g(0, "SQ_WAIT_ANY") or 0) / wc, 1)
    la = g(1, "SQ_LDS_IDX_ACTIVE")
    if la:
        e["lds_bank_conflict_pct"] = round(100 * (g(1, "SQ_LDS_BANK_CONFLICT") or 0) / la, 1)
    f, w = g(2, "FETCH_SIZE"), g(3, "WRITE_SIZE")
    if f is not None and w is not None:
        e["hbm_bytes_per_launch"] = (2 * f + w) * 1024
        e["hbm_GBps"] = round((2 * f + w) * 1024 / avg / 1e3, 1)
    res[name] = e
json.dump(res, open(os.path.join(out, "train_step_pmc.json"), "w"), indent=1)
for k, v in res.items():
    print(k, v)
