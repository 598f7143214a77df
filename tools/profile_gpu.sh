#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + PMC passes of the bench workload.
# Output under gpurun_out/prof_<tag>/ ; summaries are copied into profiles/ by hand.
# PMC passes serialise every dispatch, so they run ONE step (2 incl. the roofline leg).
set -u
TAG=${1:-r1}
STREAMS=${2:-2}          # the headline runs 2 streams of 2 clips; eager multi-stream launches keep their queues under rocprofv3
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
TRACE_ARGS="bench.py --steps 5 --warmup 2 --no-graph --no-cpu-baseline --no-full-path --no-uvg --no-train-step --no-roofline-leg --prewarm-s 0 --streams $STREAMS"
PMC_ARGS="bench.py --steps 1 --warmup 0 --no-graph --no-cpu-baseline --no-full-path --no-uvg --no-train-step --no-roofline-leg --prewarm-s 0 --streams $STREAMS"
echo "streams=$STREAMS launch=eager (--no-graph) steps_trace=5 steps_pmc=1" > $OUT/config.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $TRACE_ARGS > $OUT/trace.log 2>&1
python3 tools/prof_summary.py $OUT/trace > $OUT/kernel_trace_summary.txt 2>&1
tail -1 $OUT/trace.log > $OUT/bench_line_under_trace.json
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc1 -- python3 $PMC_ARGS > $OUT/pmc1.log 2>&1
timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d $OUT/pmc2 -- python3 $PMC_ARGS > $OUT/pmc2.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -- python3 $PMC_ARGS > $OUT/pmc3.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc4 -- python3 $PMC_ARGS > $OUT/pmc4.log 2>&1
timeout 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc5 -- python3 $PMC_ARGS > $OUT/pmc5.log 2>&1
for p in pmc1 pmc2 pmc3 pmc4 pmc5; do python3 tools/prof_summary.py $OUT/$p --pmc > $OUT/${p}_summary.txt 2>&1; done
cat $OUT/kernel_trace_summary.txt
# keep raw CSVs out of the merge-back (64 MiB cap)
find $OUT -name "*.csv" -size +1M -delete
