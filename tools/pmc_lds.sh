#!/bin/bash
OUT=gpurun_out/prof_lds; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d $OUT/pmc2 -- python3 bench.py --steps 1 --warmup 0 --no-graph --no-cpu-baseline --no-full-path --no-uvg --no-train-step --streams 1 > $OUT/pmc2.log 2>&1
python3 tools/prof_summary.py $OUT/pmc2 --pmc > $OUT/pmc2_summary.txt 2>&1
grep -A5 "fused" $OUT/pmc2_summary.txt
find $OUT -name "*.csv" -size +1M -delete
