"""gpurun_out/c3_stamps.bin (tools/experiments/c3_stamps.py) -> the launches of the last replayed step in time order: in-kernel spans
(us after the launch's first workgroup entry) and the dead time to the previous launch's end."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/c3_stamps.bin', dtype=np.uint64).reshape(256, 512, 8).astype(np.int64)
te = a[:, :, 0]; last = te.max()
L = []
for li in range(256):
    m = te[li] > 0
    if m.sum() < 8:
        continue
    t0 = te[li][m].min()
    if last - t0 > 100e6 * 0.0125:
        continue
    x = a[li][m]
    rows = bool((x[:, 3] >= x[:, 2]).all())
    end = x[:, 5].max() if rows else x[:, 2].max()
    L.append(dict(t0=t0, slot=li, wgs=int(m.sum()), ns=int(x[:, 6].max()) if rows else -1, spread=(x[:, 0].max() - t0) / 100,
                  s0=(np.median(x[:, 1] - x[:, 0])) / 100, st=(np.median(x[:, 2] - x[:, 1])) / 100,
                  ex=(np.median(x[:, 3] - x[:, 2])) / 100 if rows else 0, sto=(np.median(x[:, 4] - x[:, 3])) / 100 if rows else 0,
                  ack=(np.median(x[:, 5] - x[:, 4])) / 100 if rows else 0, end=(end - t0) / 100))
L.sort(key=lambda d: d["t0"])
prev = None
print("  # slot wgs  ns | entry spread | per workgroup (median): prologue  stages  exchange  stores  ack | launch span | gap before")
for i, d in enumerate(L):
    gap = (d["t0"] - prev) / 100 if prev else 0.0
    d["gap"] = gap
    if i < int(sys.argv[2]) if len(sys.argv) > 2 else 40:
        print("%3d %4d %3d %3d | %6.2f | %5.2f %5.2f %5.2f %5.2f %5.2f | %6.2f | %7.2f" % (i, d["slot"], d["wgs"], d["ns"], d["spread"], d["s0"], d["st"], d["ex"], d["sto"], d["ack"], d["end"], gap))
    prev = d["t0"] + d["end"] * 100
for ns in sorted(set(d["ns"] for d in L)):
    S = [d for d in L if d["ns"] == ns and d["spread"] < 2]
    if S:
        print("ns %2d one-round launches %3d: prologue %.2f stages %.2f exchange %.2f stores %.2f ack %.2f span %.2f gap %.2f" % (
            ns, len(S), *[np.mean([d[k] for d in S]) for k in ("s0", "st", "ex", "sto", "ack", "end")], np.median([d["gap"] for d in S])))
S = [d for d in L if d["spread"] >= 2]
print("launches whose workgroups entered over >= 2 us (second round):", len(S), "mean span %.2f" % (np.mean([d["end"] for d in S]) if S else 0))
print("sum of spans %.1f us over %d launches" % (sum(d["end"] for d in L), len(L)))
