"""Time the training step (config 3 of BASELINE.json: 144x144 crops, 7 frames, batch 8 per step) on one MI355X.

    python tools/bench_train.py [--batch 8] [--size 144] [--steps 10] [--warmup 3] [--fh-loss gmm]

Prints one JSON line: septuplets/s through RescaleTrainer.optimize_parameters (forward + quantise + STP + reverse +
backward + clip + Adam), ms/step, and the per-class kernel time from the library's HIP-event profiler."""
import argparse
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch  # noqa: E402


def run(batch=8, size=144, steps=10, warmup=3, fh_loss="gmm", profile=False, graph=False):
    """Time RescaleTrainer.optimize_parameters on one GPU; returns the result dict."""
    a = argparse.Namespace(batch=batch, size=size, steps=steps, warmup=warmup, fh_loss=fh_loss, profile=profile, graph=graph)
    from selfc_amd import GlobalVar, _lib, train
    from selfc_amd.modules.SelfC_GMM_arch_inv import SelfCInvNet
    dev = torch.device("cuda:0")
    GlobalVar.set_Temporal_LEN(7)
    torch.manual_seed(10)
    opt = {"global_module": "nonlocal", "stp_blk_num": 6, "fh_loss": a.fh_loss, "scale": 4, "gmm_k": 5}
    net = SelfCInvNet(opt, 3, 3, "D2DTNet", [4, 4], 2).to(dev)
    tr = train.RescaleTrainer(net, dict(train.TRAIN_OPT_LARGE), capturable=a.graph)
    g = torch.Generator().manual_seed(1234)
    gt = torch.rand(a.batch, 3, 7, a.size, a.size, generator=g).to(dev)
    real_h, ref_l, _ = train.feed_data(gt, "sr_bd", 4)
    for _ in range(a.warmup):
        tr.optimize_parameters(real_h, ref_l)
    graph_nodes = None
    if a.graph:
        from selfc_amd import runtime as rt
        keep, n_log = rt.KEEP_GRAPHS, len(rt.GRAPH_LOG)
        rt.KEEP_GRAPHS = True                    # node census of the captured step (selfc_graph_stats)
        try:
            tr.capture(real_h, ref_l)
        finally:
            rt.KEEP_GRAPHS = keep
        if len(rt.GRAPH_LOG) > n_log:
            graph_nodes = rt.GRAPH_LOG[-1]
            del rt.GRAPH_LOG[n_log:]
        if os.environ.get("SELFC_BT_TRACE"):
            print("static loss tensors", [(t_.data_ptr(), tuple(t_.shape), t_.dtype) for t_ in tr._static_losses], file=sys.stderr, flush=True)
        tr.optimize_parameters(real_h, ref_l)
    torch.cuda.synchronize()
    L = _lib.lib()
    if a.profile:
        L.selfc_profile_reset()
        L.selfc_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        log = tr.optimize_parameters(real_h, ref_l)
        if os.environ.get("SELFC_BT_TRACE"):
            print("step loss", log["loss"], log["l_forw_fit"], log["l_back_rec"], file=sys.stderr, flush=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    out = {"metric": "training septuplets/s (optimize_parameters, 7x3x%dx%d crops)" % (a.size, a.size), "value": a.batch / dt,
           "ms_per_step": dt * 1e3, "batch": a.batch, "fh_loss": a.fh_loss, "loss": log["loss"], "dtype": _lib.OPERAND,
           "launch": "hipGraph replay of the whole step" if a.graph else "eager", "graph_nodes": graph_nodes}
    if a.profile:
        L.selfc_profile_enable(0)
        names = {0: "conv3x3", 1: "conv5_F", 2: "conv5_GH", 3: "transforms", 4: "conv5_plain", 5: "stp", 6: "fused_gh", 7: "backward"}
        km = {}
        for cls, name in names.items():
            ms, n = C.c_double(), C.c_longlong()
            L.selfc_profile_read(cls, C.byref(ms), C.byref(n))
            km[name] = {"ms_per_step": ms.value / a.steps, "launches_per_step": n.value / a.steps}
        out["kernel_ms"] = km
    # the trainer dies inside a reference cycle (optimizer <-> LR scheduler): drop its hipGraphs and collect NOW, while the runtime is
    # there - left to the interpreter's final collection their destructors crashed at exit (selfc_amd/runtime.py _shutdown)
    import gc
    tr.graph = tr.graph_tail = None
    del tr, net
    gc.collect()
    torch.cuda.synchronize()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=144)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--fh-loss", default="gmm")
    ap.add_argument("--profile", action="store_true", help="also report per-class kernel time (adds event overhead)")
    ap.add_argument("--graph", action="store_true", help="capture the whole step into a hipGraph and time replays")
    a = ap.parse_args()
    print(json.dumps(run(a.batch, a.size, a.steps, a.warmup, a.fh_loss, a.profile, a.graph)))


if __name__ == "__main__":
    main()
