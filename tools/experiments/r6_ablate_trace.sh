export TMPDIR=/tmp
mkdir -p gpurun_out/r6abltr
for abl in 0 8; do
SELFC_LIB=selfc_amd/lib_dev.so SELFC_ABLATE=$abl timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r6abltr/tr$abl -- python3 tools/bench_train.py --steps 8 --warmup 3 --batch 8 --graph > gpurun_out/r6abltr/tr$abl.log 2>&1
python3 tools/trace_steps.py gpurun_out/r6abltr/tr$abl 4 > gpurun_out/r6abltr/steps_abl$abl.txt 2>&1
head -14 gpurun_out/r6abltr/steps_abl$abl.txt | cut -c1-120
done
find gpurun_out/r6abltr -name "*.csv" -size +20M -delete
