#!/bin/bash
# repeat the driver's command and print the parity leg of every run (looking for a run-to-run drift of the parity figures)
N=${1:-6}
for i in $(seq 1 $N); do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); p=d['parity']; print('run $i', d['value'], d['cpu_baseline']['cores'], round(d['cpu_baseline']['value'],3), d['cpu_baseline'].get('cpu_model'), round(p['fwd_latent_rel_err'],6), round(p['inv_rel_err'],6), p.get('debug'), 'recopied', d['cpu_baseline'].get('weights_recopied_after_a_failed_copy_check'), flush=True)" || exit 1
done
