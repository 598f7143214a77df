# re-check of the launch-geometry knobs on the final kernels (same box): least tiles per weight-gradient workgroup, chain thresholds
mkdir -p gpurun_out/r6kf
run() { # name B env...
  local name=$1 B=$2; shift 2
  env "$@" timeout -k 10 200 python3 tools/bench_train.py --steps 20 --warmup 5 --batch $B --graph > gpurun_out/r6kf/${name}_b$B.log 2>&1
  echo "$name B=$B $(tail -1 gpurun_out/r6kf/${name}_b$B.log | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d.get('ms_per_step'))")"
}
for B in 1 2 4; do
run base $B X=1
for t in 8 12 16 24; do run tiles$t $B SELFC_WG_TILES=$t; done
run cm2_512 $B SELFC_BWD_CHAIN_MAX2=512
run cm2_512_t16 $B SELFC_BWD_CHAIN_MAX2=512 SELFC_WG_TILES=16
done
run base 8 X=1
run tiles24 8 SELFC_WG_TILES=24
run cm2_1024 8 SELFC_BWD_CHAIN_MAX2=1024
run cm1_512_cm2_1024 8 SELFC_BWD_CHAIN_MAX1=512 SELFC_BWD_CHAIN_MAX2=1024
run base2 8 X=1
