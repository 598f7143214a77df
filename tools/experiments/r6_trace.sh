export TMPDIR=/tmp
O=gpurun_out/r6trace
mkdir -p $O
for B in 1 8; do
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr$B -- python3 tools/bench_train.py --steps 8 --warmup 3 --batch $B --graph > $O/tr$B.log 2>&1
python3 tools/trace_steps.py $O/tr$B 4 > $O/steps_b$B.txt 2>&1
head -48 $O/steps_b$B.txt
done
find $O -name "*.csv" -size +30M -delete
