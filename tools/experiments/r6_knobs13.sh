export TMPDIR=/tmp
O=gpurun_out/r6knobs13
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_backward.py -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
run() { tag=$1; shift; env "$@" timeout -k 10 200 python3 tools/bench_train.py --steps 30 --warmup 3 --batch $B --graph > $O/$tag.log 2>&1; python3 - <<PY
import json
try:
    d=json.loads(open("$O/$tag.log").read().strip().splitlines()[-1]); print("$tag", "B=$B", round(d["ms_per_step"],3), (d.get("graph_nodes") or {}).get("nodes"), round(d["loss"],1))
except Exception as e: print("$tag FAILED", e)
PY
}
for B in 1 8; do
  run new_b$B X=1
  run new2_b$B X=1
done
B=8
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr$B -- python3 tools/bench_train.py --steps 8 --warmup 3 --batch $B --graph > $O/tr$B.log 2>&1
python3 tools/trace_steps.py $O/tr$B 4 > $O/steps_b$B.txt 2>&1
head -44 $O/steps_b$B.txt
find $O -name "*.csv" -size +20M -delete
