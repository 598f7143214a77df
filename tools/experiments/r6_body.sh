mkdir -p gpurun_out/r6body
( time timeout -k 10 1400 python3 -m pytest tests -x -q -m gpu ) > gpurun_out/r6body/pytest.log 2>&1; tail -6 gpurun_out/r6body/pytest.log
timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6body/bench_line.json 2> gpurun_out/r6body/bench_err.log; python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r6body/bench_line.json').read().strip().splitlines()[-1])
r=d["roofline"]
print(d["value"], r["frac"], {k:r[k] for k in r if k.startswith(("train_step_ms","uvg","full_test","headline_through"))})
PY
