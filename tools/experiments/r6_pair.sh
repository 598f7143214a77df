set -e
export TMPDIR=/tmp
O=gpurun_out/r6pair
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_backward.py -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
for B in 1 8; do
  timeout -k 10 200 python3 tools/bench_train.py --steps 20 --warmup 3 --batch $B --graph > $O/bt$B.log 2>&1 || { tail -20 $O/bt$B.log; exit 1; }
  tail -1 $O/bt$B.log | cut -c1-300
  SELFC_BWD_PAIR=0 SELFC_BWD_DEFER_FIN=0 timeout -k 10 200 python3 tools/bench_train.py --steps 20 --warmup 3 --batch $B --graph > $O/bt${B}_old.log 2>&1
  tail -1 $O/bt${B}_old.log | cut -c1-300
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr$B -- python3 tools/bench_train.py --steps 8 --warmup 3 --batch $B --graph > $O/tr$B.log 2>&1
  python3 tools/trace_steps.py $O/tr$B 4 > $O/steps_b$B.txt 2>&1
  head -30 $O/steps_b$B.txt
done
find $O -name "*.csv" -size +30M -delete
