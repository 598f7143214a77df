#!/bin/bash
# headline under different stream counts / persistent-grid sizes (same box, one after the other)
B="python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-full-path --no-uvg --no-train-step"
run() { "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$TAGX', d['value'], d['ms_per_step'])"; }
for st in 1 2 4; do TAGX="streams=$st"; run timeout -k 10 120 $B --streams $st || exit 1; done
for mw in 64 96 128; do for fw in 128 192 256; do
  TAGX="gh_maxwg=$mw f_maxwg=$fw"; SELFC_FUSEDGH_MAXWG=$mw SELFC_FUSEDF_MAXWG=$fw run timeout -k 10 120 $B || exit 1
done; done
