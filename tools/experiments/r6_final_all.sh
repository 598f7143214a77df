# the round's closing sequence on one box: GPU suite, profiles (tools/experiments/r6_finalA.sh), then the driver's bench command
mkdir -p gpurun_out/r6final
( time timeout -k 10 1400 python3 -m pytest tests -x -q -m gpu ) > gpurun_out/r6final/pytest.log 2>&1; tail -6 gpurun_out/r6final/pytest.log
bash tools/experiments/r6_finalA.sh > gpurun_out/r6final/profiles.log 2>&1; tail -3 gpurun_out/r6final/profiles.log
for B in 1 2 4 8; do
timeout -k 10 200 python3 tools/bench_train.py --steps 20 --warmup 5 --batch $B --graph > gpurun_out/r6final/train_b$B.log 2>&1
echo "B=$B $(tail -1 gpurun_out/r6final/train_b$B.log | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d.get('ms_per_step'), d.get('graph_nodes'))")"
done
