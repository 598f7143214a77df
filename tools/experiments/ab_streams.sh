set -u
for r in 1 2; do for st in 1 2 4; do
python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-full-path --no-uvg --no-train-step --streams $st 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('streams $st round $r', d['value'], d['ms_per_step'], d['config']['launch'], d['box_calibration']['shader_clock_GHz_under_the_workload'])"
done; done
