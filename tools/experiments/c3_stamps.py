"""Phase stamps of the data-gradient conv launches inside the replayed training step (dev build only):
    tools/build_variant.sh dev "-DSELFC_DEV"
    SELFC_LIB=selfc_amd/lib_dev.so SELFC_ABLATE=512 python3 tools/experiments/c3_stamps.py [batch]
Per launch of the last replayed step: when (us after the launch's first workgroup entered) the median / last workgroup had its first
stage staged, finished its stages, had its output rows ready, had issued its stores, had them acknowledged."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench_train  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
r = bench_train.run(batch=B, steps=6, warmup=3, graph=True)
from selfc_amd import _lib  # noqa: E402
L = _lib.lib()
path = "gpurun_out/c3_stamps.bin"
n = L.selfc_dev_c3_stamps(path.encode())
print("ms_per_step", r["ms_per_step"], "launch counter", n)
a = np.fromfile(path, dtype=np.uint64).reshape(256, 512, 8).astype(np.int64)
t_entry = a[:, :, 0]
used = (t_entry > 0)
last = t_entry.max()
rows = []
for li in range(256):
    m = used[li]
    if m.sum() < 8:
        continue
    t0 = t_entry[li][m].min()
    if last - t0 > 100e6 * 0.020:      # older than 20 ms before the last stamp: an eager warm-up launch
        continue
    rel = (a[li][m][:, :6] - t0) / 100.0     # us
    has_rows = a[li][m][:, 3] > 0
    rows.append((t0, li, int(m.sum()), int(a[li][m][:, 6].max()), rel, has_rows))
rows.sort(key=lambda x: x[0])
names = ["entry", "stage0", "stages", "rows", "stored", "acked"]
print("launch  wgs  nstages | median (last) us after the launch's first entry: " + " ".join(names))
agg = {}
for t0, li, nw, ns, rel, hr in rows:
    if not hr.all():
        continue                      # plain-output launch: no row epilogue
    med = np.median(rel, axis=0); mx = rel.max(axis=0)
    print("%4d %5d %3d | " % (li, nw, ns) + "  ".join("%5.2f (%5.2f)" % (med[i], mx[i]) for i in range(6)))
    agg.setdefault(ns, []).append((med, mx))
for ns, v in sorted(agg.items()):
    med = np.mean([x[0] for x in v], axis=0); mx = np.mean([x[1] for x in v], axis=0)
    print("nstages %d: %3d launches | " % (ns, len(v)) + "  ".join("%5.2f (%5.2f)" % (med[i], mx[i]) for i in range(6)))
