mkdir -p gpurun_out/r6final
( time timeout -k 10 1400 python3 -m pytest tests -x -q -m gpu --durations=15 ) > gpurun_out/r6final/pytest.log 2>&1; tail -25 gpurun_out/r6final/pytest.log
timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6final/bench_line.json 2> gpurun_out/r6final/bench_err.log; tail -c 1500 gpurun_out/r6final/bench_line.json
cp gpurun_out/bench_full_line.json gpurun_out/r6final/bench_full_line.json
SELFC_BENCH_SHARE_GPU=1 timeout -k 10 600 python3 bench.py --gpus 4 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r6final/share4.json 2> gpurun_out/r6final/share4_err.log; tail -c 600 gpurun_out/r6final/share4.json
