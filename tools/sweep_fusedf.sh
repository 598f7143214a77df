#!/bin/bash
# developer sweep: bash tools/sweep_fusedf.sh "<streams list>" "<maxwg list>"
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --no-full-path --no-uvg --no-train-step"
for st in ${1:-4}; do
 for mw in ${2:-256}; do
  SELFC_FUSEDF_MAXWG=$mw timeout -k 10 120 $B --streams $st > gpurun_out/sw_${st}_${mw}.log 2>&1 || exit 1
  python - <<P
import json
d=json.loads(open('gpurun_out/sw_${st}_${mw}.log').read().strip().splitlines()[-1])
print('streams',${st},'maxwg',${mw},d['value'],d['ms_per_step'],d['kernel_ms_per_step'])
P
 done
done
