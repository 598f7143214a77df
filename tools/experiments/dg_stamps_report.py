"""gpurun_out/c3_stamps.bin -> the dgrad_chain launches (stamp 6 >= 100) of the last replayed step: per workgroup (median) the time
of the prologue (dpre4 + first fragments staged), of the layers dpre3 / dpre2 / dpre1, of the dx layers; launch span and the dead
time before the launch."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/c3_stamps.bin', dtype=np.uint64).reshape(256, 512, 8).astype(np.int64)
te = a[:, :, 0]; last = te.max()
L = []
for li in range(256):
    m = te[li] > 0
    if m.sum() < 4:
        continue
    x = a[li][m]
    t0 = x[:, 0].min()
    if last - t0 > 100e6 * 0.006:
        continue
    chain = bool((x[:, 6] >= 100).all())
    end = x[:, 5].max() if (x[:, 5] >= x[:, 2]).all() else x[:, 2].max()
    d = dict(t0=t0, slot=li, wgs=int(m.sum()), chain=chain, ns=int(x[:, 6].max()), spread=(x[:, 0].max() - t0) / 100, end=(end - t0) / 100)
    if chain:
        for k, (i, j) in dict(pro=(0, 1), l3=(1, 2), l2=(2, 3), l1=(3, 4), dx=(4, 5)).items():
            d[k] = np.median(x[:, j] - x[:, i]) / 100
    L.append(d)
L.sort(key=lambda d: d["t0"])
prev = None
print("  # slot wgs kind | entry spread | prologue  dpre3  dpre2  dpre1  dx+end | span | gap before")
for i, d in enumerate(L):
    gap = (d["t0"] - prev) / 100 if prev else 0.0
    d["gap"] = gap
    if d["chain"] and i < 400:
        print("%3d %4d %3d st%3d | %5.2f | %5.2f %5.2f %5.2f %5.2f %5.2f | %6.2f | %7.2f" % (i, d["slot"], d["wgs"], d["ns"] - 100, d["spread"], d["pro"], d["l3"], d["l2"], d["l1"], d["dx"], d["end"], gap))
    prev = d["t0"] + d["end"] * 100
C = [d for d in L if d["chain"]]
for ns in sorted(set(d["ns"] for d in C)):
    S = [d for d in C if d["ns"] == ns]
    print("stages %2d: %3d launches | prologue %.2f dpre3 %.2f dpre2 %.2f dpre1 %.2f dx+end %.2f | span %.2f" % (
        ns - 100, len(S), *[np.mean([d[k] for d in S]) for k in ("pro", "l3", "l2", "l1", "dx", "end")]))
