#!/bin/bash
# A/B of library builds on ONE box: alternates `bench.py` runs with SELFC_LIB pointing at each given .so
# (rounds interleaved, so box-to-box and clock drift cancel).  usage: tools/ab_libs.sh ROUNDS lib1.so lib2.so ...
# Prints value / fused_gh / conv3x3 (fused F) / conv5_GH ms per step for every run.
set -u
ROUNDS=$1; shift
mkdir -p gpurun_out
ARGS="bench.py --steps ${STEPS:-40} --warmup 10 --no-cpu-baseline --no-full-path --no-uvg --no-train-step --full-line"
for r in $(seq 1 $ROUNDS); do
  for lib in "$@"; do
    SELFC_LIB=$PWD/$lib timeout -k 10 120 python3 $ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); k=d['kernel_ms_per_step']
print('$lib', 'round $r', 'value', d['value'], 'ms', d['ms_per_step'], 'gh', k.get('fused_gh'), 'f', k.get('conv3x3'), 'c5gh', k.get('conv5_GH'), 'c5f', k.get('conv5_F'), 'clk', d['box_calibration']['shader_clock_GHz_under_the_workload'], 'probe', d['config'].get('graph_form_probe'))
" || exit 1
  done
done
