#!/bin/bash
# PMC passes over one training step (runs on the GPU box via gpurun; counters only - no trace domains).
export TMPDIR=/tmp
OUT=gpurun_out/pmc_train
rm -rf $OUT; mkdir -p $OUT
ARGS="tools/bench_train.py --steps 1 --warmup 1 --batch ${1:-8}"
timeout 500 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $OUT/p1 -- python3 $ARGS > $OUT/p1.log 2>&1
timeout 500 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/p2 -- python3 $ARGS > $OUT/p2.log 2>&1
python3 tools/prof_summary.py $OUT/p1 --pmc > $OUT/pmc1_summary.txt 2>&1
python3 tools/prof_summary.py $OUT/p2 --pmc > $OUT/pmc2_summary.txt 2>&1
grep -A8 -E "^wgrad_kernel|^conv3x3_kernel<12, 16, 3, 2, 4, true>|^tconv5_kernel<1, 2" $OUT/pmc1_summary.txt
grep -A7 -E "^wgrad_kernel|^conv3x3_kernel<12, 16, 3, 2, 4, true>" $OUT/pmc2_summary.txt
find $OUT -name "*.csv" -size +1M -delete
