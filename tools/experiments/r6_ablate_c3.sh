# where a data-gradient conv launch spends its time: timing-only ablations of conv3x3_kernel (-DSELFC_DEV build), captured B=8 step
mkdir -p gpurun_out/r6abl
for abl in ${ABLS:-0 1 2 4 8 6 7 15 0}; do
SELFC_LIB=selfc_amd/lib_dev.so SELFC_ABLATE=$abl timeout -k 10 200 python3 tools/bench_train.py --steps 20 --warmup 5 --batch 8 --graph > gpurun_out/r6abl/abl$abl.log 2>&1
echo "ablate=$abl $(tail -1 gpurun_out/r6abl/abl$abl.log | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d.get('ms_per_step'))")"
done
