#!/usr/bin/env python3
"""Headline benchmark: septuplets/s (7x3x256x448), forward + inverse InvBlock stack.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: one rank per GPU over RCCL.  Without WORLD_SIZE in the environment bench.py starts its own N ranks -
     a child `python -m torch.distributed.run --nproc-per-node N bench.py ...`, before this process touches the
     GPU - and relays rank 0's JSON line and exit code; under an existing torch.distributed.run it simply is a rank.
     --dry-run rehearses launch / sharding / timing protocol on CPU over gloo without any HIP call.)

One step = the hot path over one batch of 4 synthetic septuplets per GPU, already
resident in HBM: FrequencyAnalyzer.fwd -> 8 x InvBlockExp.fwd -> Quantization ->
8 x InvBlockExp.rev -> FrequencyAnalyzer.rev (BASELINE.json configs[1]).
Septuplets are independent: ranks shard them with no data-path collective (weak
scaling); the only collectives are the timing barrier and the MAX over ranks.
Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline`
(dense 3x3 conv kernel, MFMA-bound, timed live with HIP events on its launch
stream) and `cpu_baseline` (the CPU oracle = a port of the reference path, timed
on this box's host cores at N=1).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

T, H, W, B_PER_GPU = 7, 256, 448, 4
OPT = {"global_module": "nonlocal", "stp_blk_num": 6, "fh_loss": "gmm", "scale": 4, "gmm_k": 5}
PEAK_F16_TFLOPS = 2500.0     # dense f16/bf16 MFMA peak, MI355X_MICROARCH.md "Chip-level parameters"
# algorithmic MACs per LR pixel-frame per block and direction
MAC_F14_PX = 9 * 32 * (48 + 80 + 112 + 144)          # conv1..4 of F   (4 conv3x3_kernel launches)
MAC_GH14_PX = 2 * 9 * 32 * (3 + 35 + 67 + 99)        # conv1..4 of G+H (1 fused_gh_kernel launch)
MAC_BLOCK_PX = 267408                                # whole InvBlockExp


def build_net(device):
    from selfc_amd import GlobalVar
    from selfc_amd.modules.SelfC_GMM_arch_inv import SelfCInvNet
    GlobalVar.set_Temporal_LEN(T)
    torch.manual_seed(10)                      # manual_seed of train_rescaling_selfc_large.yml
    net = SelfCInvNet(OPT, 3, 3, "D2DTNet", [4, 4], 2).eval()
    return net.to(device)


_HASH_MUL = 0x9E3779B97F4A7C15 - (1 << 64)          # odd 64-bit multiplier (as a signed int64)


def tensor_hash(t: torch.Tensor) -> int:
    """Position-dependent 64-bit hash of a float32 tensor's BITS, computed where the tensor lives (device tensors: on the device,
    only the 8-byte result crosses PCIe): sum over i of (bits[i] + 1) * (2 i + 1) * M, wrapping in int64.  Unlike a sum of |w| it
    moves with a permutation, a sign flip or a swapped pair of elements; the same function on a host copy must give the same value."""
    b = t.detach().contiguous().view(torch.int32).flatten().to(torch.int64)
    i = torch.arange(b.numel(), dtype=torch.int64, device=b.device)
    return int((((b + 1) * (2 * i + 1)) * _HASH_MUL).sum().item())


def weights_hash(net) -> dict:
    """tensor name -> tensor_hash of the device tensor, for every floating-point entry of the state dict"""
    return {k: tensor_hash(v) for k, v in net.state_dict().items() if v.is_floating_point()}


def box_identity() -> dict:
    """Which physical card / host this run landed on (sysfs only - no HIP call): makes "one box" a testable statement when a
    fault is seen on some runs of a pool and not on others (profiles/r5/host_copy_*.txt)."""
    out = {}
    try:
        import glob
        import socket
        out["host"] = socket.gethostname()
        for d in sorted(glob.glob("/sys/class/drm/card*/device")):
            try:
                if open(os.path.join(d, "vendor")).read().strip() != "0x1002":
                    continue
                uid = os.path.join(d, "unique_id")
                out.setdefault("gpus", []).append({"pci": os.path.basename(os.path.realpath(d)),
                                                   "unique_id": open(uid).read().strip() if os.path.exists(uid) else None})
            except OSError:
                continue
        out["boot_id"] = open("/proc/sys/kernel/random/boot_id").read().strip()
        if torch.cuda.is_initialized():       # which of the host's cards this process runs on (sysfs lists all of them)
            pr = torch.cuda.get_device_properties(torch.cuda.current_device())
            out["device"] = {k: (str(getattr(pr, k)) if k == "uuid" else getattr(pr, k)) for k in ("name", "pci_domain_id", "pci_bus_id", "pci_device_id", "uuid") if hasattr(pr, k)}
    except Exception as e:  # noqa: BLE001
        out["error"] = repr(e)[:100]
    return out


def describe_bad_copy(name, wrong: torch.Tensor, dev_t: torch.Tensor, others: dict) -> dict:
    """A host copy that does not match its device tensor: WHERE it differs (byte ranges), WHAT the wrong bytes are (hex sample,
    all zero?), and whether they equal the same region of any other buffer we can still name (`others`: name -> host tensor of
    the same shape, e.g. the training leg's final / initial weights).  The pair is also saved under gpurun_out/ for offline study."""
    import numpy as np
    right = dev_t.detach().cpu()
    w8 = wrong.contiguous().view(torch.uint8).flatten().numpy()
    r8 = right.contiguous().view(torch.uint8).flatten().numpy()
    bad = np.flatnonzero(w8 != r8)
    info = {"tensor": name, "nbytes": int(w8.size), "bad_bytes": int(bad.size), "second_copy_matches_the_device": bool(torch.equal(right.to(dev_t.device), dev_t))}
    if bad.size:
        cuts = np.flatnonzero(np.diff(bad) > 1)
        runs = [(int(bad[a]), int(bad[b]) + 1) for a, b in zip(np.r_[0, cuts + 1], np.r_[cuts, bad.size - 1])]
        lo, hi = int(bad[0]), int(bad[-1]) + 1
        info.update({"first_bad_byte": lo, "last_bad_byte": hi - 1, "bad_runs": runs[:16], "n_bad_runs": len(runs),
                     "span_aligned_64": [lo % 64, hi % 64], "wrong_bytes_all_zero": bool((w8[lo:hi] == 0).all()),
                     "wrong_hex_first_32": w8[lo:lo + 32].tobytes().hex(), "right_hex_first_32": r8[lo:lo + 32].tobytes().hex()})
        same = []
        for oname, o in others.items():
            try:
                o8 = o.detach().cpu().contiguous().view(torch.uint8).flatten().numpy()
                if o8.size == w8.size and (o8[lo:hi] == w8[lo:hi]).all():
                    same.append(oname)
            except Exception:  # noqa: BLE001
                continue
        info["wrong_bytes_equal_the_same_region_of"] = same
        # is the wrong span a shifted piece of the right tensor (a copy that landed at the wrong offset)?
        span = w8[lo:hi].tobytes()
        at = r8.tobytes().find(span) if hi - lo >= 16 else -1
        info["wrong_span_found_in_the_right_tensor_at_byte"] = at
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        fn = os.path.join(ROOT, "gpurun_out", f"host_copy_mismatch_{int(time.time())}_{name.replace('.', '_')}.npz")
        np.savez(fn, wrong=w8, right=r8)
        info["dump"] = os.path.relpath(fn, ROOT)
    except Exception as e:  # noqa: BLE001
        info["dump_error"] = repr(e)[:100]
    return info


def host_weights(net, others=None):
    """The oracle's copy of the weights.  Every host copy is checked against the device tensor two ways - copied back and compared
    on the device bit for bit, and tensor_hash(host copy) against tensor_hash taken on the device - and a copy that fails is NOT
    silently replaced: it is described (describe_bad_copy: offsets, contents, what else the bytes equal), dumped, and returned in
    `errors`, which the caller puts into the bench line as an error field.  The oracle is then fed a second, verified copy so that
    the parity figures stay a statement about the kernels.  Why this exists: profiles/r4/parity_leg_host_copy.txt (9 of 45 runs on
    one day got one 3456-byte weight wrong in this copy); deliberately NO device synchronize in front of the copies - `v.cpu()`
    orders against the current stream only, and a missing join with some other stream is exactly what this is meant to expose."""
    params, errors = {}, []
    for k, v in net.state_dict().items():
        if not k.startswith("operations."):
            continue
        v = v.detach()
        c = v.cpu()
        if v.is_cuda and not (torch.equal(c.to(v.device), v) and tensor_hash(c) == tensor_hash(v)):
            errors.append(describe_bad_copy(k, c, v, {f"{n}:{kk}": o[kk] for n, o in (others or {}).items() for kk in o
                                                      if kk != k and o[kk].numel() == v.numel()}))
            c = v.cpu()
            if not torch.equal(c.to(v.device), v):
                raise RuntimeError(f"bench: the host copy of {k} does not match the device tensor twice in a row")
        params[k] = c
    return params, errors


def cpu_baseline(net, x_cpu, budget_s=20.0, others=None):
    """Oracle (port of the reference's torch path) on one septuplet of the same workload.

    torch's CPU convs scale badly past a few dozen threads on a many-core host, so
    the thread count is calibrated on a 128x224 crop first and the one used is
    reported as `cores`."""
    from oracle import selfc_oracle as O      # checker / baseline only
    params, copy_errors = host_weights(net, others)
    ncpu = os.cpu_count() or 1
    small = x_cpu[:, :, :128, :224].contiguous()
    best_t, best_n = None, 1
    with torch.no_grad():
        for nthr in sorted({min(ncpu, c) for c in (8, 16, 32, 64)}):
            torch.set_num_threads(nthr)
            O.large_fwd(params, small, T)
            t0 = time.perf_counter()
            O.large_fwd(params, small, T)
            dt = time.perf_counter() - t0
            if best_t is None or dt < best_t:
                best_t, best_n = dt, nthr
        torch.set_num_threads(best_n)
        t0 = time.perf_counter()
        z = O.large_fwd(params, x_cpu, T)                     # also the parity reference
        zq = torch.cat((O.quantize(z[:, :3]), z[:, 3:]), 1)
        xr = O.large_inv_from_latent(params, zq, T)
        times = [time.perf_counter() - t0]
        while sum(times) < budget_s and len(times) < 16:          # ~10-20 s of CPU work on the GPU box's host
            t0 = time.perf_counter()
            O.large_roundtrip(params, x_cpu, T)
            times.append(time.perf_counter() - t0)
    timed = times[1:] if len(times) > 1 else times
    med = sorted(timed)[len(timed) // 2]
    try:
        cpu_model = next(ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name"))
    except Exception:  # noqa: BLE001
        cpu_model = None
    cpu_baseline.params = params                  # kept for the parity leg's self-diagnosis
    return {"value": 1.0 / med, "unit": "septuplets/s", "cores": best_n, "kind": "port", "cpu_model": cpu_model,
            "host_copy_errors": copy_errors,      # [] = every device -> host weight copy was bit-exact at the first attempt
            "sample": f"1 septuplet 7x3x{H}x{W}, fwd+quant+inv through the CPU oracle (torch fp32), {len(timed)} timed run(s), median; "
                      f"{best_n} threads (calibrated) of {ncpu} host CPUs"}, z, zq, xr


def mark(what: str):
    """progress marker on stderr (never stdout: the line is the only thing printed there): where a run was when it died.
    SELFC_BENCH_MARKS=1 only: the driver keeps the last 2,000 characters of stdout + stderr, and they belong to the line."""
    if os.environ.get("SELFC_BENCH_MARKS") == "1":
        print(f"[bench {time.strftime('%H:%M:%S')}] {what}", file=sys.stderr, flush=True)


def short(text, n=118):
    """the driver's record truncates strings at 120 characters"""
    text = str(text)
    return text if len(text) <= n else text[:n - 3] + "..."


def rocprof_reference(kernel_key: str, csrc_sha16: str):
    """avg launch duration of `kernel_key` in the TRACKED rocprofv3 kernel trace of this round (profiles/r6/rocprof_kernel_avgs.json,
    written by tools/rocprof_avgs.py from the committed trace of this same command) - next to the live HIP-event figure."""
    try:
        with open(os.path.join(ROOT, "profiles", "r6", "rocprof_kernel_avgs.json")) as fh:
            ref = json.load(fh)
        k = ref.get(kernel_key) or next((v for n_, v in ref.items() if n_ != "_meta" and kernel_key and n_.startswith(kernel_key)), None)
        meta = ref.get("_meta", {})
        if not k:
            return None
        return {"avg_us": k["avg_us"], "file": "profiles/r6/rocprof_kernel_avgs.json", "same_sources": meta.get("csrc_sha16") == csrc_sha16}
    except (OSError, ValueError, KeyError):
        return None


def compact_line(out: dict) -> dict:
    """The ONE line the driver records: its standard keys, `config`, `roofline` and `cpu_baseline` - scalars only (the record keeps
    no nested objects and no other top-level key: BENCH_r05.parsed.extra_keys), strings under 120 characters.  The secondary legs
    travel as flat scalars inside `roofline`; the verbose form is `--full-line` (and gpurun_out/bench_full_line.json)."""
    keys = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    line = {k: out.get(k) for k in keys}
    cfg = out.get("config", {})
    cal = out.get("box_calibration") or {}
    line["config"] = {"workload": "SelfC-large FreqAnalyzer + 8 InvBlockExp(D2DTNet) fwd, Quantization, 8 rev, FreqAnalyzer rev; 4 x 7x3x256x448 in HBM",
                      "septuplets_per_gpu": cfg.get("septuplets_per_gpu"), "launch": short(cfg.get("launch")), "streams": cfg.get("streams"),
                      "sharding": short(cfg.get("sharding")), "prewarm": short(cfg.get("prewarm")),
                      "rccl_ranks": out.get("rccl_ranks"), "shader_clock_GHz": cal.get("shader_clock_GHz_under_the_workload"),
                      "box_mfma_f16_loop_TFLOPs": cal.get("mfma_f16_loop_TFLOPs"), "box_device_copy_GBps": cal.get("device_copy_GBps")}
    rf = out.get("roofline")
    if rf is not None:
        r = {k: (short(v) if isinstance(v, str) else v) for k, v in rf.items() if not isinstance(v, (dict, list))}
        sr = out.get("stack_roofline") or {}
        r["stack_mfma_frac"] = sr.get("mfma_frac")
        r["stack_hbm_frac_layer_granular"] = sr.get("hbm_frac_layer_granular")
        c5 = (out.get("roofline_other") or {}).get("conv5_GH") or {}
        r["conv5_GH_hbm_GBps"], r["conv5_GH_hbm_frac"] = c5.get("achieved"), c5.get("frac")
        ff = (out.get("roofline_other") or {}).get("conv3x3") or {}
        r["fused_f_mfma_frac"] = ff.get("frac")
        ts = out.get("train_step") or {}
        by = ts.get("captured_ms_per_step_by_local_batch") or {}
        r["train_step_ms_b8"], r["train_step_ms_b4"], r["train_step_ms_b2"], r["train_step_ms_b1"] = by.get("8"), by.get("4"), by.get("2"), by.get("1")
        r["train_step_mfma_frac"], r["train_step_graph_nodes"] = ts.get("mfma_frac"), ts.get("graph_nodes")
        r["train_step_graph_nodes_b1"], r["train_step_eager_ms"] = ts.get("graph_nodes_b1"), ts.get("eager_ms_per_step")
        uv = out.get("uvg_1080p") or {}
        r["uvg_1080p_frames_per_s"], r["uvg_1080p_mfma_frac_whole_path"] = uv.get("frames_per_s"), uv.get("mfma_frac_whole_path")
        fp = out.get("full_test_path") or {}
        r["full_test_path_septuplets_per_s"] = fp.get("septuplets_per_s")
        r["full_test_path_mfma_frac_whole_path"] = fp.get("mfma_frac_whole_path")
        r["headline_through_module_api_septuplets_per_s"] = (out.get("headline_through_module_api") or {}).get("septuplets_per_s")
        pa = out.get("parity") or {}
        r["parity_fwd_rel_err"], r["parity_inv_rel_err"] = pa.get("fwd_latent_rel_err"), pa.get("inv_rel_err")
        r["parity_fwd_rel_l2"], r["parity_inv_rel_l2"], r["parity_error"] = pa.get("fwd_latent_rel_l2"), pa.get("inv_rel_l2"), short(pa.get("error")) if pa.get("error") else None
        for leg in ("train_step", "uvg_1080p", "full_test_path"):
            if isinstance(out.get(leg), dict) and out[leg].get("error"):
                r[leg + "_error"] = short(out[leg]["error"])
        line["roofline"] = {k: (round(v, 6) if isinstance(v, float) else v) for k, v in r.items()}
    else:
        line["roofline"] = None
    cb = out.get("cpu_baseline")
    if cb is not None:
        line["cpu_baseline"] = {"value": round(cb["value"], 4), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                                "cpu_model": short(cb.get("cpu_model")), "host_copy_errors": len(cb.get("host_copy_errors") or []),
                                "sample": "1 septuplet 7x3x256x448 fwd+quant+inv, CPU oracle (torch fp32), median of the timed runs, calibrated threads"}
    return line


def main():
    import faulthandler
    faulthandler.enable()          # a crash inside a native call (HIP runtime, a kernel launch) leaves the Python stack on stderr
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed steps (2.8 ms each: the rate keeps rising until ~100 steps - clocks, caches - so short runs under-report the steady state by ~4 %%)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--graph-mode", choices=("auto", "single", "per-stream"), default="auto",
                    help="multi-stream hipGraph form: one graph with a branch per stream, one graph per stream, or (auto) whichever is faster on this box - timed BEFORE the timed region, reported in config.launch")
    ap.add_argument("--prewarm-s", type=float, default=1.0, help="untimed pre-warm (seconds of steps) before the W warm-up steps and the timed region; reported in config.prewarm")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-full-path", action="store_true", help="skip the extra netG(x) + Quantization + netG(LR, rev=True) timing (incl. STP sampler)")
    ap.add_argument("--streams", type=int, default=2, help="split the septuplets of a step over this many HIP streams (2 x 2 clips: measured best since the round-2 G/H kernel; 4 x 1 clip before)")
    ap.add_argument("--no-uvg", action="store_true", help="skip the 1080p leg (config 5, bounded sample: 6 GOPs through the whole test path)")
    ap.add_argument("--no-train-step", action="store_true", help="skip the extra training-step timing (config 3: 8 x 7x3x144x144)")
    ap.add_argument("--no-roofline-leg", action="store_true", help="profiling runs only (tools/profile_gpu.sh): skip the eager one-stream leg that times every launch, so that a trace holds the timed configuration alone; `roofline` is then null")
    ap.add_argument("--full-line", action="store_true", help="print the verbose dict (every leg's details, per-kernel rooflines, box identity) instead of the compact line the driver records")
    ap.add_argument("--dry-run", action="store_true", help="CPU / gloo rehearsal of launch, sharding and the timing protocol: no HIP call, value is null")
    args = ap.parse_args()

    from selfc_amd import launch
    rc = launch.self_launch(args.gpus, os.path.abspath(__file__), sys.argv[1:])     # before anything touches the GPU
    if rc is not None:
        sys.exit(rc)
    if args.dry_run:
        return dry_run(args)
    # SELFC_BENCH_SHARE_GPU=1 (rehearsal on a one-GPU box, tests): every rank runs on cuda:0 and the protocol's collectives go over
    # gloo - the N-rank launch, sharding, barrier and max-over-ranks timing on real kernels, everything except RCCL.  The line says so.
    share_gpu = os.environ.get("SELFC_BENCH_SHARE_GPU") == "1" and args.gpus > 1
    local = 0 if share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    if local >= torch.cuda.device_count():       # device_count() does not initialise HIP
        raise SystemExit(f"rank with LOCAL_RANK={local} but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    ranks = launch.Ranks(args.gpus, "gloo" if share_gpu else "nccl", dev)
    world, rank = ranks.world, ranks.rank

    from selfc_amd import _lib
    from selfc_amd.pipeline import MultiStreamRoundTrip, RescaleRoundTrip
    L = _lib.lib()
    net = build_net(dev)
    w_hash0 = weights_hash(net)
    n_frames = B_PER_GPU * T
    g = torch.Generator().manual_seed(launch.rank_seed(1234, rank))
    x_cpu = torch.rand(n_frames, 3, H, W, generator=g)
    x = x_cpu.to(dev)
    rt = RescaleRoundTrip(net, n_frames, H, W, dev)
    runner = rt if args.streams <= 1 else MultiStreamRoundTrip(net, n_frames, H, W, dev, args.streams)

    with torch.no_grad():
        use_graph = not args.no_graph
        if use_graph:
            runner.capture(x)
            step = runner.replay
        else:
            step = lambda: runner.run(x)      # noqa: E731
        # ---- roofline leg FIRST: K steps, eager, one stream, HIP events around every launch on its stream.  Running it (and
        # the stated pre-warm below) ahead of the timed region means the timed region is not the first GPU work of the
        # process: a 20-step run then measures the same steady state as a 200-step one (clocks and caches settled).
        L.selfc_profile_reset()
        if not args.no_roofline_leg:
            L.selfc_profile_enable(1)
            for _ in range(args.steps):
                rt.run(x)
            torch.cuda.synchronize()
            L.selfc_profile_enable(0)
        cls_ms, cls_n = {}, {}
        for cls, name in [(0, "conv3x3"), (1, "conv5_F"), (2, "conv5_GH"), (3, "transforms"), (6, "fused_gh")]:
            ms, n = C.c_double(), C.c_longlong()
            L.selfc_profile_read(cls, C.byref(ms), C.byref(n))
            cls_ms[name], cls_n[name] = ms.value, n.value
        L.selfc_profile_reset()
        # ---- graph form (untimed tuning): a two-branch hipGraph runs at the one-stream rate on some boxes of the pool
        graph_form, graph_probe = "one graph", None
        if use_graph and args.streams > 1:
            def rate(n=30):
                for _ in range(10):
                    step()
                torch.cuda.synchronize()
                best = 1e9
                for _ in range(2):
                    t0 = time.perf_counter()
                    for _ in range(n):
                        step()
                    torch.cuda.synchronize()
                    best = min(best, (time.perf_counter() - t0) / n)
                return best * 1e3
            if args.graph_mode == "auto":
                t_single = rate()
                runner.capture(x, per_stream=True)
                t_per = rate()
                graph_probe = {"one_graph_ms": round(t_single, 3), "graph_per_stream_ms": round(t_per, 3)}
                if t_single <= t_per:
                    runner.capture(x)
                else:
                    graph_form = "one graph per stream"
            elif args.graph_mode == "per-stream":
                runner.capture(x, per_stream=True)
                graph_form = "one graph per stream"
        # ---- stated pre-warm: untimed steps for --prewarm-s seconds of wall time (in addition to the W warm-up steps)
        t_pw, n_pw = time.perf_counter(), 0
        while time.perf_counter() - t_pw < args.prewarm_s:
            for _ in range(10):
                step()
            torch.cuda.synchronize()
            n_pw += 10
        # ---- the timed region: W untimed warm-up steps, barrier + sync, EXACTLY K steps, barrier + sync, MAX over ranks
        dt = launch.timed_region(step, args.steps, args.warmup, ranks, torch.cuda.synchronize)
        rccl_ranks = ranks.count()
        # ---- box calibration, AFTER the timed region (boxes of one pool differ by up to 10 % on one binary): a register-only
        # MFMA loop and a 512-MiB device copy, so that a reader can tell a slow box from a slow build
        # (the headline never depends on it: a calibration that cannot run - e.g. no room for its 1-GiB copy buffer - is reported as such)
        clk_ghz, cal_m, cal_c, cal_err = None, C.c_double(0.0), C.c_double(0.0), None
        try:
            clk = torch.zeros(2, dtype=torch.int64, device=dev)
            side = torch.cuda.Stream()
            torch.cuda.synchronize()
            n_clk = max(4, min(args.steps, 40))
            _lib.check(L.selfc_profile_clock_sample(clk.data_ptr(), max(1, min(400000, int(0.8 * n_clk * dt / args.steps * 1e6))), side.cuda_stream),
                       "selfc_profile_clock_sample")
            for _ in range(n_clk):
                step()
            torch.cuda.synchronize()
            clk_ghz = 0.1 * float(clk[0]) / max(1.0, float(clk[1]))
            _lib.check(L.selfc_profile_calibrate(C.byref(cal_m), C.byref(cal_c), _lib.stream_ptr()), "selfc_profile_calibrate")
        except RuntimeError as e:
            cal_err = str(e)[:200]
            torch.cuda.synchronize()

    mark("headline timed, calibration done")
    npx = n_frames * (H // 4) * (W // 4)
    # per-kernel rooflines: algorithmic FLOPs per launch / live HIP-event duration of that launch
    kern = {}
    # class "conv3x3" = conv1-4 of F: one timed scope per block and direction when the pairwise-fused kernels run
    # (fused_f_kernel<0> + fused_f_kernel<1>, two launches under one scope), four (one per conv) on the layer-wise path
    f_scopes = max(1, round(cls_n.get("conv3x3", 0) / (16.0 * args.steps)))
    f_label = ("fused_f16_kernel<0> + <1> (conv1-4 of F, pairwise fused on v_mfma_f32_16x16x32, two launches timed as one)" if f_scopes == 1
               else "conv3x3_kernel (conv1-4 of F, layer-wise)")
    for name, mac_px, per_blockdir, label in (("conv3x3", MAC_F14_PX, f_scopes, f_label),
                                              ("fused_gh", MAC_GH14_PX, 1, "fused_gh_kernel (conv1-4 of G+H, fused)")):
        if cls_n.get(name, 0) == 0:
            continue
        fl = 2.0 * mac_px * npx / per_blockdir
        avg_ms = cls_ms[name] / cls_n[name]
        tf = fl / (avg_ms * 1e-3) / 1e12
        kern[name] = {"bound": "mfma", "kernel": label, "achieved": round(tf, 1), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                      "frac": round(tf / PEAK_F16_TFLOPS, 4), "traffic": None, "avg_launch_us": round(avg_ms * 1e3, 2),
                      "launches": cls_n[name], "flops_per_launch": fl, "ms_per_step": round(cls_ms[name] / args.steps, 3)}
    # Counter-derived figures cannot be collected inside this process (rocprofv3 wraps a whole run): they come from the
    # committed PMC passes of this same workload (tools/profile_gpu.sh -> tools/pmc_traffic.py) and sit under `pmc_reference`
    # with the configuration and commit they were taken at.  `traffic` (HBM bytes per launch) is only filled from them while the
    # kernel sources still hash to what was profiled; otherwise it stays null.
    import hashlib
    traffic_all, pmc_meta = {}, {}
    for rnd in ("r6", "r5", "r4", "r3", "r2"):
        try:
            with open(os.path.join(ROOT, "profiles", rnd, "pmc_traffic.json")) as fh:
                traffic_all = json.load(fh)
            pmc_meta = dict(traffic_all.pop("_meta", {}), file=f"profiles/{rnd}/pmc_traffic.json")
            break
        except (OSError, ValueError):
            continue
    hsh = hashlib.sha256()
    csrc = os.path.join(ROOT, "selfc_amd", "csrc")
    for f in sorted(n_ for n_ in os.listdir(csrc) if n_.endswith((".hip", ".hpp"))):      # every kernel source and header
        hsh.update(f.encode())
        hsh.update(open(os.path.join(csrc, f), "rb").read())
    same_sources = bool(pmc_meta.get("csrc_sha16")) and pmc_meta.get("csrc_sha16") == hsh.hexdigest()[:16]
    pmc_meta["kernel_sources_unchanged_since"] = same_sources

    def pmc_file_of(name):
        """path of the newest tracked counter file of that name + whether ITS OWN source hash (its _meta.csrc_sha16) is the current one
        (the flag used to compare against pmc_traffic.json's hash whatever file was meant)"""
        for rnd in ("r6", "r5"):
            path = os.path.join(ROOT, "profiles", rnd, name)
            try:
                with open(path) as fh:
                    meta = json.load(fh).get("_meta", {})
                own = meta.get("csrc_sha16")
                return {"file": f"profiles/{rnd}/{name}", "same_sources": (own == hsh.hexdigest()[:16]) if own else None}
            except (OSError, ValueError):
                continue
        return None

    def add_pmc(entry, key):
        t = traffic_all.get(key)
        if t:
            # per-launch byte counts scale with the frames of a launch: the passes ran `frames_per_launch` frames per launch,
            # the eager roofline leg above launches all B_PER_GPU * T frames at once
            scale = n_frames / float(pmc_meta.get("frames_per_launch") or n_frames)
            if same_sources:
                entry["traffic"] = t["hbm_bytes_per_launch"] * scale
            entry["pmc_reference_file"] = pmc_meta.get("file")
            entry["pmc_reference_same_sources"] = same_sources
    if f_scopes == 1 and "fused_f" in traffic_all:
        traffic_all["conv3x3"] = traffic_all["fused_f"]
    for k in kern:
        add_pmc(kern[k], k)
    # HBM-bound kernel of the path: temporal conv5 of G+H with the affine coupling fused (tconv5_kernel<2,3,4,1,3>):
    # algorithmic bytes per LR pixel-frame = 2 x 128 f16 features + 12 B y1 + 192 B x2 in, 192 B y2 + 96 B f16 copy out
    if cls_n.get("conv5_GH", 0):
        by = 1004.0 * npx
        avg_ms = cls_ms["conv5_GH"] / cls_n["conv5_GH"]
        gbs = by / (avg_ms * 1e-3) / 1e9
        kern["conv5_GH"] = {"bound": "hbm", "kernel": "tconv5_kernel<2,3,4,1,3> (temporal conv5 of G+H + affine coupling)", "achieved": round(gbs, 1),
                            "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4), "traffic": None, "avg_launch_us": round(avg_ms * 1e3, 2),
                            "launches": cls_n["conv5_GH"], "bytes_per_launch": by, "ms_per_step": round(cls_ms["conv5_GH"] / args.steps, 3)}
        add_pmc(kern["conv5_GH"], "conv5_GH")
    dominant = max(kern, key=lambda k: kern[k]["ms_per_step"]) if kern else None
    roofline = kern[dominant] if kern else None
    if roofline is not None:
        ref_ = rocprof_reference({"fused_gh": "fused_gh_kernel", "conv3x3": "fused_f16_kernel<1>", "conv5_GH": "tconv5_kernel<2, 3, 4, 1, 3"}.get(dominant, ""),
                                 hsh.hexdigest()[:16])
        # the same kernel's average in the tracked rocprofv3 trace of this command (another card, under the profiler): README explains gaps > 5 %
        roofline["rocprof_avg_us"] = ref_["avg_us"] if ref_ else None
        roofline["rocprof_file"] = ref_["file"] if ref_ else None
        roofline["rocprof_same_sources"] = ref_["same_sources"] if ref_ else None
    value = launch.whole_job_rate(B_PER_GPU, world, args.steps, dt)
    whole_flops = 2.0 * MAC_BLOCK_PX * npx * 16
    out = {
        "metric": "septuplets/sec (7x3x256x448) fwd+inv InvBlock stack", "value": round(value, 2), "unit": "septuplets/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": _lib.OPERAND, "data": "synthetic",
        "config": {"workload": "SelfC-large FrequencyAnalyzer + 8 InvBlockExp(D2DTNet) fwd, Quantization, 8 InvBlockExp rev, "
                               "FrequencyAnalyzer rev; 4 septuplets 7x3x256x448 per GPU, inputs resident in HBM, seeded default-init weights",
                   "septuplets_per_gpu": B_PER_GPU, "launch": (f"hipGraph replay ({graph_form})" if args.streams > 1 else "hipGraph replay") if use_graph else "eager", "streams": args.streams,
                   "graph_form_probe": graph_probe,
                   "prewarm": f"untimed, before the W warm-up steps: the eager roofline leg ({args.steps} steps) + {n_pw} steps over {args.prewarm_s} s of wall time",
                   "sharding": f"{world} rank(s) x {B_PER_GPU} independent septuplets, no data-path collective"
                               + (" - REHEARSAL: all ranks share ONE GPU (SELFC_BENCH_SHARE_GPU=1), collectives over gloo; not a scaling figure" if share_gpu else "")},
        "rccl_ranks": rccl_ranks,
        "box_identity": box_identity(),
        "box_calibration": {"shader_clock_GHz_under_the_workload": None if clk_ghz is None else round(clk_ghz, 3),
                            "mfma_f16_loop_TFLOPs": round(cal_m.value, 1) or None, "device_copy_GBps": round(cal_c.value, 1) or None, "error": cal_err,
                            "what": "rank 0, right after the timed region: shader clock sampled by one wave on a side stream over more steps of the same workload "
                                    "(shader-clock counter against the 100 MHz counter); register-only 32x32x16 f16 MFMA loop; 512-MiB D2D copy, read + write bytes"},
        "roofline": roofline,
        "roofline_other": {k: v for k, v in kern.items() if k != dominant},
        "stack_tflops": round(whole_flops * world * args.steps / dt / 1e12, 1),
        # SURVEY 8d reporting rule: the whole unit of work against both rooflines, per GPU.  Layer-granular byte model =
        # 1,815 16-bit elements per pixel-frame per block and direction (one kernel per conv on a concat-free buffer),
        # 2.91 GB per septuplet; the north star's ">= 50 % of the HBM roofline" is this fraction.
        "stack_roofline": {"mfma_frac": round(whole_flops * args.steps / dt / 1e12 / PEAK_F16_TFLOPS, 4),
                           "hbm_frac_layer_granular": round(2 * 8 * 1815 * 2 * npx * args.steps / dt / 8.0e12, 4),
                           "hbm_peak_GBps": 8000, "bytes_model": "2*8*1815 f16 elements per LR pixel-frame (SURVEY 8d)"},
        # eager, one stream, HIP events around every launch: conv3x3 = conv1-4 of F, conv5_F = conv5 of F + coupling
        # (f_couple_kernel on the fused path), fused_gh = conv1-4 of G+H, conv5_GH = conv5 of G+H + coupling
        "kernel_ms_per_step": {k: round(v / args.steps, 3) for k, v in cls_ms.items()},
    }
    if not args.no_full_path and world == 1:
        # SelfCModel.test()'s two netG calls through the MODULE API (SelfC_model.py:213-230): netG(x) -> Quantization -> netG(LR, rev=True),
        # NCHW fp32 in, fresh NCHW tensors out, STP + GMM sampler on the reverse.  In eval / no_grad the modules route through their
        # cached two-stream hipGraph (pipeline.ModuleGraph) from the second call of a shape on - this is what a maintainer who makes
        # the three import changes of INTEGRATION.md runs.
        from selfc_amd import pipeline as PL
        from selfc_amd.modules.Quantization import Quantization
        quant = Quantization()

        def time_calls(fn, n_warm, n_timed):
            for _ in range(n_warm):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n_timed):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n_timed
        with torch.no_grad():
            def full():
                z, _ = net(x=x, rev=False)
                return net(x=quant(z[:, :3]), rev=True)[0]

            def stack_only():          # the headline's unit of work through the module API: STP bypassed, the forward's own HF channels fed back
                z, _ = net(x=x, rev=False)
                return net.inverse_from_latent(torch.cat((quant(z[:, :3]), z[:, 3:]), 1))
            tf = time_calls(full, 5, 40)
            ts = time_calls(stack_only, 5, 40)
            # the same calls on the eager single-stream launches (first-use / fallback path), with HIP events around every launch
            PL.MODULE_GRAPH = False
            te = time_calls(full, 2, 5)
            L.selfc_profile_reset()
            L.selfc_profile_enable(1)
            for _ in range(5):
                full()
            torch.cuda.synchronize()
            L.selfc_profile_enable(0)
            PL.MODULE_GRAPH = True
            fp = {}
            for cls, name in [(0, "conv3x3"), (1, "conv5_F"), (2, "conv5_GH"), (3, "transforms"), (4, "conv5_plain"), (5, "stp"), (6, "fused_gh")]:
                ms, n = C.c_double(), C.c_longlong()
                L.selfc_profile_read(cls, C.byref(ms), C.byref(n))
                fp[name] = round(ms.value / 5, 3)
            L.selfc_profile_reset()
        out["full_test_path"] = {"septuplets_per_s": round(B_PER_GPU / tf, 1), "ms_per_batch": round(tf * 1e3, 3),
                                 # stack + STP algorithmic FLOPs (546.4 GFLOP per septuplet) over the measured time
                                 "mfma_frac_whole_path": round((2.0 * MAC_BLOCK_PX * 16 + 2331776.0) * npx / tf / 1e12 / PEAK_F16_TFLOPS, 4),
                                 "note": "module API: netG(x=x) -> Quantization -> netG(x=LR, rev=True), fresh NCHW tensors in and out, fh_loss gmm with device RNG; "
                                         "each call replays its cached two-stream hipGraph (pipeline.ModuleGraph), 40 timed batches after 5",
                                 "eager": {"septuplets_per_s": round(B_PER_GPU / te, 1), "ms_per_batch": round(te * 1e3, 3), "kernel_ms": fp,
                                           "note": "the same calls with SELFC_MODULE_GRAPH=0: eager launches on one stream (the first-use / fallback path)"}}
        out["headline_through_module_api"] = {"septuplets_per_s": round(B_PER_GPU / ts, 1), "ms_per_batch": round(ts * 1e3, 3),
                                              "note": "the headline's unit of work through the drop-in calls: netG(x=x) -> Quantization of the LR channels -> "
                                                      "netG.inverse_from_latent(cat(LR, the forward's own HF)): STP BYPASSED as in the headline; includes the NCHW "
                                                      "conversions, the cat and the fresh output tensors the headline's pre-bound pipeline does not pay"}
        # the same work as a pre-bound pipeline: hipGraph, one septuplet per stream (pipeline.FullTestPath)
        try:
            from selfc_amd.pipeline import FullTestPath
            with torch.no_grad():
                ftp = MultiStreamRoundTrip(net, n_frames, H, W, dev, args.streams if args.streams > 1 else 1, part_cls=FullTestPath)
                ftp.capture(x)
                tg = time_calls(ftp.replay, 10, 100)
            out["full_test_path"]["pipeline"] = {"septuplets_per_s": round(B_PER_GPU / tg, 1), "ms_per_batch": round(tg * 1e3, 3),
                                                  "launch": f"hipGraph replay, {ftp.nstreams} streams"}
            out["full_test_path"]["module_over_pipeline"] = round(tg / tf, 4)
            del ftp
        except Exception as e:  # noqa: BLE001
            out["full_test_path"]["pipeline"] = {"error": repr(e)[:300]}
    mark("full test path leg done")
    if not args.no_uvg and world == 1:
        # config 5 of BASELINE.json, bounded sample: 1080p GOPs (7x3x1080x1920) through the whole test path, two GOPs per hipGraph
        # replay on two streams (tools/bench_uvg.py measure(); the full 100-frame clips / --gpus N live there).  Secondary leg.
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        try:
            import bench_uvg
            torch.cuda.empty_cache()
            frames = 42                                                   # 6 GOPs = 3 replays
            sec, ngop = bench_uvg.measure(net, dev, 1, frames, 1080, 1920, 2, ranks)
            out["uvg_1080p"] = dict({"frames_per_s": round(frames / sec, 1), "gops_per_s": round(ngop / sec, 2), "seconds": round(sec, 4),
                                     "sample": f"{ngop} GOPs of 7x3x1080x1920 (one synthetic {frames}-frame clip), whole test path (fwd stack, Quantization, STP sample, rev stack), "
                                               "2 GOPs per hipGraph replay on 2 streams, incl. the device copies of each GOP into the graph's input"},
                                    **bench_uvg.roofline_fracs(ngop, 1, sec, 1080, 1920))
            out["uvg_1080p"]["pmc_reference_file"] = pmc_file_of("uvg1080p_pmc.json")
            torch.cuda.empty_cache()
        except Exception as e:  # noqa: BLE001
            out["uvg_1080p"] = {"error": repr(e)[:300]}
    mark("1080p leg done")
    if not args.no_train_step and world == 1:
        # config 3 of BASELINE.json: one optimize_parameters step (fwd, quantise, STP sample, reverse, backward, clip, Adam)
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
        try:                                          # a secondary leg must never cost the headline line
            import bench_train
            tg = bench_train.run(batch=8, size=144, steps=20, warmup=3, fh_loss="gmm", profile=False, graph=True)
            te = bench_train.run(batch=8, size=144, steps=20, warmup=3, fh_loss="gmm", profile=False)
            by_batch, nodes_b1 = {}, None
            for lb in (1, 2, 4):       # the reference's global batch 8 split over 8 / 4 / 2 ranks (train_rescaling_selfc_large.yml:12,26: 2 GPUs x 4)
                r_ = bench_train.run(batch=lb, size=144, steps=20, warmup=2, fh_loss="gmm", profile=False, graph=True)
                by_batch[str(lb)] = round(r_["ms_per_step"], 2)
                if lb == 1:
                    nodes_b1 = (r_.get("graph_nodes") or {}).get("nodes")
            by_batch["8"] = round(tg["ms_per_step"], 2)
            # algorithmic work of a training step: forward + data gradient + weight gradient of the whole test path (stack fwd + rev,
            # STP) = 3 x 10.89 MFLOP per LR pixel-frame, 36x36 latent pixels x 7 frames per septuplet
            step_flop = lambda lb: 3.0 * (2.0 * MAC_BLOCK_PX * 16 + 2331776.0) * lb * T * 36 * 36      # noqa: E731
            out["train_step"] = {"septuplets_per_s": round(tg["value"], 1), "ms_per_step": round(tg["ms_per_step"], 2),
                                 "captured_ms_per_step_by_local_batch": by_batch,
                                 "graph_nodes": (tg.get("graph_nodes") or {}).get("nodes"), "graph_nodes_b1": nodes_b1,
                                 "mfma_frac_by_local_batch": {k_: round(step_flop(int(k_)) / (v_ * 1e-3) / 1e12 / PEAK_F16_TFLOPS, 4) for k_, v_ in by_batch.items()},
                                 "mfma_frac": round(step_flop(8) / (tg["ms_per_step"] * 1e-3) / 1e12 / PEAK_F16_TFLOPS, 4),
                                 "launch": "RescaleTrainer.capture(): the whole optimisation step replayed as one hipGraph",
                                 "eager_ms_per_step": round(te["ms_per_step"], 2), "eager_septuplets_per_s": round(te["value"], 1),
                                 "config": "8 x 7x3x144x144 crops, fh_loss gmm, l2 + l1 losses, clip 10, Adam (one flat tensor); 3 streams",
                                 "note": "every forward / reverse / gradient kernel is HIP (selfc_amd/autograd.py); losses, clip and Adam are torch; "
                                         "the eager figure is host-bound and moves with the box's CPU"}
            out["train_step"]["pmc_reference_file"] = pmc_file_of("train_step_pmc.json")
        except Exception as e:  # noqa: BLE001
            out["train_step"] = {"error": repr(e)[:300]}
    mark("training leg done")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cb, z_ref, zq_ref, xr_ref = cpu_baseline(net, x_cpu[:T], others={"this_net": net.state_dict()})
        out["cpu_baseline"] = cb
        with torch.no_grad():
            z = rt.forward_latent(x)[:T].cpu()
            # inverse half on the SAME quantised latent as the oracle (a comparison through the
            # quantiser would turn a 1e-4 LR difference into a full 1/255 step)
            zin = torch.zeros(n_frames, 51, H // 4, W // 4)
            zin[:T] = zq_ref
            xr = rt.inverse_latent(zin.to(dev))[:T].cpu()
        def mx(a, b):
            return float((a - b).abs().max() / b.abs().max())

        def l2(a, b):
            return float((a.double() - b.double()).norm() / b.double().norm())
        out["parity"] = {"fwd_latent_rel_err": mx(z, z_ref), "inv_rel_err": mx(xr, xr_ref),
                         # two-sided: relative L2, and the LR (0:3) / HF (3:51) channel groups each against their own magnitude
                         "fwd_latent_rel_l2": l2(z, z_ref), "inv_rel_l2": l2(xr, xr_ref),
                         "fwd_lr_rel_err": mx(z[:, :3], z_ref[:, :3]), "fwd_hf_rel_err": mx(z[:, 3:], z_ref[:, 3:]),
                         "fwd_lr_rel_l2": l2(z[:, :3], z_ref[:, :3]), "fwd_hf_rel_l2": l2(z[:, 3:], z_ref[:, 3:]),
                         "tolerance": 1e-3, "metric": "max|a-b|/max|b| (rel_err), ||a-b||/||b|| (rel_l2)", "against": "CPU oracle, septuplet 0"}
        # the device weights must still be the ones the net was built with (per-tensor position-dependent hash taken on the device)
        changed = [k for k, h_ in weights_hash(net).items() if w_hash0.get(k) != h_]
        out["parity"]["device_weights_changed_since_build"] = changed[:8]
        if cb["host_copy_errors"] or changed:
            out["parity"]["error"] = ("a device -> host weight copy did not match its device tensor at the first attempt (cpu_baseline.host_copy_errors: "
                                      "offsets, contents, dump); the oracle was fed a second, verified copy" if cb["host_copy_errors"] else
                                      "device weights changed during the run")
        if max(out["parity"]["fwd_latent_rel_err"], out["parity"]["inv_rel_err"]) > 1e-3:
            # never expected: say which side moved (device result repeated, a freshly packed runner, the drop-in module call,
            # the oracle recomputed on one thread, the weights' checksum against the one taken when the net was built)
            dbg = {}
            try:
                from oracle import selfc_oracle as O      # checker only
                with torch.no_grad():
                    dbg["fwd_again"] = mx(rt.forward_latent(x)[:T].cpu(), z_ref)
                    rt2 = RescaleRoundTrip(net, T, H, W, dev)
                    dbg["fwd_fresh_runner_one_septuplet"] = mx(rt2.forward_latent(x[:T].contiguous()).cpu(), z_ref)
                    zm, _ = net(x=x[:T].contiguous(), rev=False)
                    dbg["fwd_module_call"] = mx(zm.cpu(), z_ref)
                    p0 = cpu_baseline.params
                    dbg["oracle_again_same_threads_same_params_vs_first"] = mx(O.large_fwd(p0, x_cpu[:T], T), z_ref)
                    params, _ = host_weights(net)
                    dbg["params_of_the_first_pass_differ_in"] = [k for k in params if not torch.equal(params[k], p0[k])][:8]
                    dbg["oracle_again_same_threads_new_params_vs_first"] = mx(O.large_fwd(params, x_cpu[:T], T), z_ref)
                    torch.set_num_threads(1)
                    dbg["oracle_one_thread_vs_first"] = mx(O.large_fwd(params, x_cpu[:T], T), z_ref)
                dbg["torch_threads_of_the_reference"] = cb["cores"]
            except Exception as e:  # noqa: BLE001
                dbg["error"] = repr(e)[:300]
            out["parity"]["debug"] = dbg
    mark("cpu baseline / parity leg done")
    if rank == 0:
        print(json.dumps(out if args.full_line else compact_line(out)), flush=True)
        try:                                   # the verbose record, best effort (scratch on the GPU box, merged back by gpurun)
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "bench_full_line.json"), "w") as fh:
                json.dump(out, fh)
        except OSError:
            pass
    mark("line printed")
    ranks.close()
    # leave nothing to interpreter finalisation: the training legs' trainers die inside reference cycles (optimizer <-> scheduler), and a
    # hipGraph destroyed by the FINAL garbage collection segfaulted in 2 of 14 runs after the line was printed (runtime._shutdown)
    import gc
    del runner, rt, net
    gc.collect()
    torch.cuda.synchronize()
    mark("main() returns")


def dry_run(args):
    """Launch / shard / timing protocol on CPU over gloo (tests, and a rehearsal of --gpus N where there is no GPU): the
    step is a placeholder, so the line says "dry_run": true and carries no throughput."""
    from selfc_amd import launch
    ranks = launch.Ranks(args.gpus, "gloo")
    g = torch.Generator().manual_seed(launch.rank_seed(1234, ranks.rank))
    x = torch.rand(B_PER_GPU * T, 3, 8, 8, generator=g)          # this rank's septuplets (tiny stand-ins)
    sink = []
    dt = launch.timed_region(lambda: sink.append(float(x.sum())), args.steps, args.warmup, ranks)
    n = ranks.count()
    sums = [None] * ranks.world
    if ranks.dist is not None:
        ranks.dist.all_gather_object(sums, sink[-1])
    else:
        sums = [sink[-1]]
    if ranks.rank == 0:
        print(json.dumps({"metric": "septuplets/sec (7x3x256x448) fwd+inv InvBlock stack", "value": None, "unit": "septuplets/s",
                          "dry_run": True, "n_gpus": ranks.world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(dt / max(1, args.steps) * 1e3, 6), "scaling": "weak", "rccl_ranks": n,
                          "backend": "gloo", "septuplets_per_gpu": B_PER_GPU, "shards_distinct": len(set(sums)) == len(sums),
                          "steps_counted": len(sink) - args.warmup}), flush=True)
    ranks.close()


if __name__ == "__main__":
    main()
