export TMPDIR=/tmp
bash tools/profile_gpu.sh r6_1s 1 > gpurun_out/prof1_1s.log 2>&1
bash tools/profile_gpu.sh r6_2s 2 > gpurun_out/prof1_2s.log 2>&1
bash tools/pmc_train.sh 8 > gpurun_out/prof2_train8.log 2>&1
mkdir -p gpurun_out/r6trace_final
for B in 1 8; do
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r6trace_final/tr$B -- python3 tools/bench_train.py --steps 8 --warmup 3 --batch $B --graph > gpurun_out/r6trace_final/tr$B.log 2>&1
python3 tools/trace_steps.py gpurun_out/r6trace_final/tr$B 4 > gpurun_out/r6trace_final/steps_b$B.txt 2>&1
done
head -6 gpurun_out/r6trace_final/steps_b1.txt; head -6 gpurun_out/r6trace_final/steps_b8.txt
bash tools/profile_uvg.sh r6 > gpurun_out/prof_uvg.log 2>&1; tail -5 gpurun_out/prof_uvg.log
find gpurun_out/r6trace_final gpurun_out/pmc_train gpurun_out/prof_uvg_r6 -name "*.csv" -size +20M -delete
