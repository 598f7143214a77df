mkdir -p gpurun_out/r6rowepi
timeout -k 10 900 python3 -m pytest tests/test_gpu_backward.py tests/test_gpu_train.py -x -q -m gpu > gpurun_out/r6rowepi/pytest.log 2>&1; tail -3 gpurun_out/r6rowepi/pytest.log
for B in 8 4 1 8; do
timeout -k 10 200 python3 tools/bench_train.py --steps 20 --warmup 5 --batch $B --graph > gpurun_out/r6rowepi/b$B.log 2>&1
echo "B=$B $(tail -1 gpurun_out/r6rowepi/b$B.log | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d.get('ms_per_step'))")"
done
SELFC_LIB=selfc_amd/lib_dev.so SELFC_ABLATE=512 timeout -k 10 300 python3 tools/experiments/c3_stamps.py 8 > gpurun_out/c3_stamps_b8.txt 2>&1; tail -6 gpurun_out/c3_stamps_b8.txt
