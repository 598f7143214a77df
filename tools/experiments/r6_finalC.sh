mkdir -p gpurun_out/r6final
timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6final/bench_line.json 2> gpurun_out/r6final/bench_err.log; tail -c 2500 gpurun_out/r6final/bench_line.json
cp gpurun_out/bench_full_line.json gpurun_out/r6final/bench_full_line.json
