"""What clock do the fused kernels run at INSIDE the benchmark's hipGraph?  (diagnostic build only)

    make -C selfc_amd/csrc diag DIAG=-DSELFC_CLOCKS && SELFC_LIB=selfc_amd/libselfc_diag.so python tools/clock_probe.py

Thread 0 of workgroup 0 of every fused F / fused G/H launch adds its shader-cycle count and its 100 MHz tick count to a
slot (csrc/common.hpp ClockProbe); after a few seconds of back-to-back graph replays (the headline configuration) this prints,
per kernel, the mean cycles of that workgroup and the clock = cycles / ticks x 100 MHz (MI355X_MICROARCH.md, DVFS give-back 6)."""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    from selfc_amd import _lib
    from selfc_amd.pipeline import MultiStreamRoundTrip, RescaleRoundTrip
    L = _lib.lib()
    if not hasattr(L, "selfc_debug_clocks"):
        raise SystemExit("load the diagnostic library: make -C selfc_amd/csrc diag DIAG=-DSELFC_CLOCKS; SELFC_LIB=selfc_amd/libselfc_diag.so")
    dev = torch.device("cuda:0")
    net = bench.build_net(dev)
    n = bench.B_PER_GPU * bench.T
    x = torch.rand(n, 3, bench.H, bench.W, generator=torch.Generator().manual_seed(1234)).to(dev)
    buf = (C.c_ulonglong * 24)()
    out = {}
    for streams in (2, 1):
        with torch.no_grad():
            r = RescaleRoundTrip(net, n, bench.H, bench.W, dev) if streams == 1 else MultiStreamRoundTrip(net, n, bench.H, bench.W, dev, streams)
            r.capture(x)
            for _ in range(50):
                r.replay()
            torch.cuda.synchronize()
            L.selfc_debug_clocks(buf, 1)
            t0, k = time.perf_counter(), 0
            while time.perf_counter() - t0 < 3.0:
                for _ in range(20):
                    r.replay()
                torch.cuda.synchronize()
                k += 20
            dt = time.perf_counter() - t0
        assert L.selfc_debug_clocks(buf, 1) == 0
        res = {"steps": k, "ms_per_step": round(dt / k * 1e3, 3), "septuplets_per_s": round(bench.B_PER_GPU * k / dt, 1)}
        for slot, name in ((0, "fused_f<0>"), (1, "fused_f<1>"), (2, "fused_gh")):
            cyc, ticks, cnt = buf[3 * slot], buf[3 * slot + 1], buf[3 * slot + 2]
            if cnt:
                res[name] = {"launches": cnt, "cycles_wg0": round(cyc / cnt), "us_wg0": round(ticks / cnt / 100.0, 2), "clock_GHz": round(cyc / ticks * 0.1, 3)}
        out[f"streams_{streams}_graph"] = res
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
