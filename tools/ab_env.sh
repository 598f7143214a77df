#!/bin/bash
# A/B of environment switches on ONE box: alternates short `bench.py` runs (driver mode: --steps 20 --warmup 5) under each
# given environment assignment (rounds interleaved, so box-to-box and clock drift cancel).
#   usage: tools/ab_env.sh ROUNDS "VAR=1" "" "OTHER=x SELFC_LIB=..."        ("" = default environment)
set -u
ROUNDS=$1; shift
mkdir -p gpurun_out
ARGS="bench.py --steps ${STEPS:-20} --warmup ${WARMUP:-5} --no-cpu-baseline --no-full-path --no-uvg --no-train-step --full-line ${EXTRA:-}"
for r in $(seq 1 $ROUNDS); do
  for v in "$@"; do
    env $v timeout -k 10 180 python3 $ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); k=d['kernel_ms_per_step']
print('[$v]', 'round $r', 'value', d['value'], 'ms', d['ms_per_step'], 'gh', k.get('fused_gh'), 'f', k.get('conv3x3'), 'c5gh', k.get('conv5_GH'), 'c5f', k.get('conv5_F'), 'clk', d['box_calibration']['shader_clock_GHz_under_the_workload'], 'par', d.get('parity'), flush=True)
" || exit 1
  done
done
