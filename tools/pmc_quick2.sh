#!/bin/bash
# one PMC pass (SQ busy / MFMA busy / waits) of bench.py --streams 1, eager; prints the fused kernels' rows.  Extra env applies to the run.
OUT=gpurun_out/prof_q2; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc1 -- python3 bench.py --steps 1 --warmup 0 --no-graph --no-cpu-baseline --no-full-path --no-uvg --no-train-step --no-roofline-leg --prewarm-s 0 --streams 1 > $OUT/pmc1.log 2>&1
python3 tools/prof_summary.py $OUT/pmc1 --pmc > $OUT/pmc1_summary.txt 2>&1
grep -A9 "fused_gh\|fused_f_kernel<1>" $OUT/pmc1_summary.txt
find $OUT -name "*.csv" -size +1M -delete
