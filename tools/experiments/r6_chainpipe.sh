mkdir -p gpurun_out/r6cp
timeout -k 10 900 python3 -m pytest tests/test_gpu_backward.py tests/test_gpu_train.py -x -q -m gpu > gpurun_out/r6cp/pytest.log 2>&1; tail -3 gpurun_out/r6cp/pytest.log
for B in 1 2 4 8 1; do
timeout -k 10 200 python3 tools/bench_train.py --steps 20 --warmup 5 --batch $B --graph > gpurun_out/r6cp/b$B.log 2>&1
echo "B=$B $(tail -1 gpurun_out/r6cp/b$B.log | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d.get('ms_per_step'))")"
done
