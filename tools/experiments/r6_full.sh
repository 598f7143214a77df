export TMPDIR=/tmp
O=gpurun_out/r6full
mkdir -p $O
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
run() { tag=$1; shift; env "$@" timeout -k 10 200 python3 tools/bench_train.py --steps 30 --warmup 3 --batch $B --graph > $O/$tag.log 2>&1; python3 - <<PY
import json
try:
    d=json.loads(open("$O/$tag.log").read().strip().splitlines()[-1]); print("$tag", "B=$B", round(d["ms_per_step"],3), (d.get("graph_nodes") or {}).get("nodes"), round(d["loss"],1))
except Exception as e: print("$tag FAILED", e)
PY
}
for B in 1 2 4 8; do
  run base_b$B X=1
  run fp80k_b$B SELFC_T5_FP_MAX=80000
  run fp40k_b$B SELFC_T5_FP_MAX=40000
done
