#!/bin/bash
# Kernel trace + PMC passes over the training step (config 3: 8 x 7x3x144x144, eager; runs on the GPU box via gpurun; counter passes carry
# no trace domains).  python3 sits directly behind `--`.  tools/pmc_train.py turns the summaries into profiles/rN/train_step_pmc.json.
export TMPDIR=/tmp
OUT=gpurun_out/pmc_train
rm -rf $OUT; mkdir -p $OUT
B=${1:-8}
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/bench_train.py --steps 6 --warmup 2 --batch $B > $OUT/trace.log 2>&1
python3 tools/prof_summary.py $OUT/trace > $OUT/kernel_trace_summary.txt 2>&1
ARGS="tools/bench_train.py --steps 1 --warmup 1 --batch $B"
timeout 500 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $OUT/p1 -- python3 $ARGS > $OUT/p1.log 2>&1
timeout 500 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/p2 -- python3 $ARGS > $OUT/p2.log 2>&1
timeout 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/p3 -- python3 $ARGS > $OUT/p3.log 2>&1
timeout 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/p4 -- python3 $ARGS > $OUT/p4.log 2>&1
for p in 1 2 3 4; do python3 tools/prof_summary.py $OUT/p$p --pmc > $OUT/pmc${p}_summary.txt 2>&1; done
head -14 $OUT/kernel_trace_summary.txt
find $OUT -name "*.csv" -size +1M -delete
