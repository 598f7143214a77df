#!/bin/bash
# A/B of library builds on the captured training step (tools/bench_train.py --graph), local batch 8 / 4 / 1, rounds interleaved.
# usage: tools/ab_train.sh ROUNDS lib1.so lib2.so ...
set -u
ROUNDS=$1; shift
for r in $(seq 1 $ROUNDS); do for b in ${BATCHES:-8 4 1}; do for lib in "$@"; do
  SELFC_LIB=$PWD/$lib timeout -k 10 200 python3 tools/bench_train.py --batch $b --steps 30 --warmup 2 --graph 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$lib', 'round $r', 'batch', d['batch'], 'ms', round(d['ms_per_step'],3), 'loss', round(d['loss'],1), flush=True)" || exit 1
done; done; done
