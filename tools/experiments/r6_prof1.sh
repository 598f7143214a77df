export TMPDIR=/tmp
bash tools/profile_gpu.sh r6_1s 1 > gpurun_out/prof1_1s.log 2>&1
bash tools/profile_gpu.sh r6_2s 2 > gpurun_out/prof1_2s.log 2>&1
tail -25 gpurun_out/prof1_2s.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6_bench_line.json 2> gpurun_out/r6_bench_err.txt; echo "bench rc $?"
cat gpurun_out/r6_bench_line.json | cut -c1-3000
