export TMPDIR=/tmp
bash tools/pmc_train.sh 8 > gpurun_out/prof2_train8.log 2>&1
tail -16 gpurun_out/prof2_train8.log
mkdir -p gpurun_out/r6trace_final
for B in 1 8; do
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r6trace_final/tr$B -- python3 tools/bench_train.py --steps 8 --warmup 3 --batch $B --graph > gpurun_out/r6trace_final/tr$B.log 2>&1
python3 tools/trace_steps.py gpurun_out/r6trace_final/tr$B 4 > gpurun_out/r6trace_final/steps_b$B.txt 2>&1
done
head -12 gpurun_out/r6trace_final/steps_b1.txt
SELFC_BENCH_SHARE_GPU=1 timeout -k 10 600 python3 bench.py --gpus 6 --steps 10 --warmup 3 > gpurun_out/r6_share_gpu_6.json 2> gpurun_out/r6_share_gpu_6.err; echo "share rc $?"; tail -c 600 gpurun_out/r6_share_gpu_6.json
timeout -k 10 900 python -m pytest tests/test_gpu_data.py tests/test_gpu_backward.py -q --durations=12 > gpurun_out/r6_data_tests.log 2>&1; tail -22 gpurun_out/r6_data_tests.log
find gpurun_out/r6trace_final gpurun_out/pmc_train -name "*.csv" -size +20M -delete
