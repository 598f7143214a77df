#!/bin/bash
# Repeat the driver's command and print, per run: the headline, the parity leg, the box identity (the card the process runs on: PCI bus
# id / uuid from the device properties, the number of cards the host lists, boot id of the host: "one box" as a testable statement)
# and any device -> host weight copy that failed its check (bench.py host_weights: offsets, contents and a dump under gpurun_out/ -
# no silent retry).  A run that prints no line keeps its stderr (gpurun_out/parity_repeat_fail_<i>.err).  profiles/r5/host_copy_repeats.txt
# collects the output.
N=${1:-6}
mkdir -p gpurun_out
for i in $(seq 1 $N); do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --full-line > gpurun_out/_pr_line.json 2> gpurun_out/_pr_err.txt; rc=$?
  if [ $rc -ne 0 ] || [ ! -s gpurun_out/_pr_line.json ]; then
    cp gpurun_out/_pr_err.txt gpurun_out/parity_repeat_fail_$i.err
    echo "run $i FAILED rc=$rc (line on stdout: $(wc -c < gpurun_out/_pr_line.json) bytes): $(grep -v amdgpu.ids gpurun_out/_pr_err.txt | grep -v '^Extension modules' | tail -12 | tr '\n' '|' | cut -c1-900)"
    if [ ! -s gpurun_out/_pr_line.json ]; then continue; fi
  fi
  python3 -c "
import json
d=json.loads(open('gpurun_out/_pr_line.json').read().strip().splitlines()[-1]); p=d['parity']; b=d.get('box_identity',{})
print('run $i', d['value'], 'clk', d['box_calibration']['shader_clock_GHz_under_the_workload'], 'box', b.get('host'), (b.get('device') or {}).get('pci_bus_id'), (b.get('device') or {}).get('uuid'), len(b.get('gpus') or []), 'cards', b.get('boot_id','')[:8],
      'parity', round(p['fwd_latent_rel_err'],6), round(p['inv_rel_err'],6), 'error', p.get('error'), 'host_copy_errors', d['cpu_baseline'].get('host_copy_errors'), p.get('debug'), flush=True)"
done
