set -e
export TMPDIR=/tmp
mkdir -p gpurun_out/r6base
python -m pytest tests -m gpu -x -q > gpurun_out/r6base/pytest.log 2>&1 || { tail -30 gpurun_out/r6base/pytest.log; exit 1; }
tail -3 gpurun_out/r6base/pytest.log
python bench.py > gpurun_out/r6base/bench.log 2>gpurun_out/r6base/bench.err
tail -c 3000 gpurun_out/r6base/bench.log
for B in 1 8; do
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r6base/tr$B -- python3 tools/bench_train.py --steps 8 --warmup 3 --batch $B --graph > gpurun_out/r6base/tr$B.log 2>&1
  python3 tools/trace_queues.py gpurun_out/r6base/tr$B 7 3 14 > gpurun_out/r6base/queues_b$B.txt 2>&1
  tail -1 gpurun_out/r6base/tr$B.log
done
find gpurun_out/r6base -name "*.csv" -size +30M -delete
