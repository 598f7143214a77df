#!/bin/bash
# Repeat the driver's command and print, per run: the headline, the parity leg, the box identity (PCI id / unique id of the card,
# boot id of the host: "one box" as a testable statement) and any device -> host weight copy that failed its check (bench.py
# host_weights: offsets, contents and a dump under gpurun_out/ - no silent retry).  profiles/r5/host_copy_*.txt collects the output.
N=${1:-6}
for i in $(seq 1 $N); do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); p=d['parity']; b=d.get('box_identity',{}); g=(b.get('gpus') or [{}])[0]
print('run $i', d['value'], 'clk', d['box_calibration']['shader_clock_GHz_under_the_workload'], 'box', b.get('host'), (b.get('device') or {}).get('pci_bus_id'), (b.get('device') or {}).get('uuid'), len(b.get('gpus') or []), 'cards', b.get('boot_id','')[:8],
      'parity', round(p['fwd_latent_rel_err'],6), round(p['inv_rel_err'],6), 'error', p.get('error'), 'host_copy_errors', d['cpu_baseline'].get('host_copy_errors'), p.get('debug'), flush=True)" || exit 1
done
