export TMPDIR=/tmp
O=gpurun_out/r6knobs14
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_backward.py -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
run() { tag=$1; shift; env "$@" timeout -k 10 200 python3 tools/bench_train.py --steps 30 --warmup 3 --batch $B --graph > $O/$tag.log 2>&1; python3 - <<PY
import json
try:
    d=json.loads(open("$O/$tag.log").read().strip().splitlines()[-1]); print("$tag", "B=$B", round(d["ms_per_step"],3), (d.get("graph_nodes") or {}).get("nodes"), round(d["loss"],1))
except Exception as e: print("$tag FAILED", e)
PY
}
for B in 1 2 4 8; do
  run new_b$B X=1
  run perplane_b$B SELFC_T5B_ALL=0
  run new2_b$B X=1
done
