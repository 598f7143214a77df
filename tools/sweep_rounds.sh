#!/bin/bash
# headline vs the number of frames every persistent G/H workgroup walks (more rounds = fewer, longer-lived workgroups per launch);
# interleaved repetitions on one box
B="python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-full-path --no-uvg --no-train-step"
for rep in 1 2 3; do for gr in 2 3 4; do
  SELFC_FUSEDGH_MINROUNDS=$gr timeout -k 10 120 $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('rep $rep GH minrounds $gr', d['value'], d['ms_per_step'])" || exit 1
done; done
