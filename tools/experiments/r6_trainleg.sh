mkdir -p gpurun_out/r6leg
for i in 1 2; do timeout -k 10 200 python3 tools/bench_train.py --steps 20 --warmup 3 --batch 8 --graph > gpurun_out/r6leg/a$i.log 2>&1; echo "standalone $i: $(tail -1 gpurun_out/r6leg/a$i.log | python3 -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])")"; done
timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r6leg/bench.json 2>/dev/null; python3 -c "
import json; d=json.loads(open('gpurun_out/r6leg/bench.json').read().strip().splitlines()[-1]); print('bench leg', d['roofline']['train_step_ms_b8'], d['value'])"
timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-uvg --no-full-path > gpurun_out/r6leg/bench2.json 2>/dev/null; python3 -c "
import json; d=json.loads(open('gpurun_out/r6leg/bench2.json').read().strip().splitlines()[-1]); print('bench leg without uvg/full path', d['roofline']['train_step_ms_b8'], d['value'])"
timeout -k 10 200 python3 tools/bench_train.py --steps 20 --warmup 3 --batch 8 --graph > gpurun_out/r6leg/a3.log 2>&1; echo "standalone 3: $(tail -1 gpurun_out/r6leg/a3.log | python3 -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])")"
