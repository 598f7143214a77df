#!/bin/bash
# developer sweep of the persistent kernels' launch geometry in the 4-stream bench
B="python bench.py --no-cpu-baseline --no-full-path --no-uvg --no-train-step --streams 4"
for fr in 1 2 3 4; do for gr in 1 2 3 4; do
  SELFC_FUSEDF_MINROUNDS=$fr SELFC_FUSEDGH_MINROUNDS=$gr timeout -k 10 120 $B > gpurun_out/sg.log 2>&1 || exit 1
  python - <<P
import json
d=json.loads(open('gpurun_out/sg.log').read().strip().splitlines()[-1])
print('F minrounds',$fr,'GH minrounds',$gr,d['value'],d['ms_per_step'])
P
done; done
