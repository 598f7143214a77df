#!/bin/bash
# which leg / library makes bench.py's parity leg drift?  prints the parity dict per configuration
p() { python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('$1', d['value'], {k: round(v,6) for k,v in d['parity'].items() if k in ('fwd_latent_rel_err','inv_rel_err')}, flush=True)"; }
python3 bench.py --steps 5 --warmup 2 --no-full-path --no-uvg --no-train-step 2>/dev/null | p "current, headline only"
SELFC_LIB=$PWD/tools/experiments/lib_prev.so python3 bench.py --steps 5 --warmup 2 --no-full-path --no-uvg --no-train-step 2>/dev/null | p "lib_prev, headline only"
python3 bench.py --steps 5 --warmup 2 --no-uvg --no-train-step 2>/dev/null | p "current, + full path"
python3 bench.py --steps 5 --warmup 2 --no-full-path --no-train-step 2>/dev/null | p "current, + uvg"
python3 bench.py --steps 5 --warmup 2 --no-full-path --no-uvg 2>/dev/null | p "current, + train"
