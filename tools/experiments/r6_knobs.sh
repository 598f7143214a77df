export TMPDIR=/tmp
O=gpurun_out/r6knobs
mkdir -p $O
run() { tag=$1; shift; env "$@" timeout -k 10 200 python3 tools/bench_train.py --steps 20 --warmup 3 --batch $B --graph > $O/$tag.log 2>&1; python3 - <<PY
import json
try:
    d=json.loads(open("$O/$tag.log").read().strip().splitlines()[-1]); print("$tag", "B=$B", round(d["ms_per_step"],3), (d.get("graph_nodes") or {}).get("nodes"))
except Exception as e: print("$tag FAILED", e)
PY
}
for B in 1 8; do
  run base_b$B X=1
  run chain1_b$B SELFC_BWD_CHAIN=1
  run chain0_b$B SELFC_BWD_CHAIN=0
  run onestream_b$B SELFC_BWD_STREAMS=1
  run onestream_chain1_b$B SELFC_BWD_STREAMS=1 SELFC_BWD_CHAIN=1
  run nodefer_b$B SELFC_BWD_DEFER_FIN=0
  run old_b$B SELFC_BWD_PAIR=0 SELFC_BWD_DEFER_FIN=0
done
