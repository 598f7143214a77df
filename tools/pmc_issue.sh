#!/bin/bash
# Runs on the GPU box: where the waves' cycles go, per kernel - instruction-issue activity by unit, instruction fetch,
# outstanding-request levels.  Output: gpurun_out/pmc_issue/{a,b,c}_summary.txt
set -u
OUT=gpurun_out/pmc_issue
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="bench.py --steps 1 --warmup 0 --no-graph --no-cpu-baseline --no-full-path --no-uvg --no-train-step --streams 1"
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/a -- python3 $ARGS > $OUT/a.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $OUT/b -- python3 $ARGS > $OUT/b.log 2>&1
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_LDS_DATA_FIFO_FULL --output-format csv -d $OUT/c -- python3 $ARGS > $OUT/c.log 2>&1
for p in a b c; do python3 tools/prof_summary.py $OUT/$p --pmc > $OUT/${p}_summary.txt 2>&1; done
find $OUT -name "*.csv" -size +1M -delete
for p in a b c; do grep -A9 "fused_gh_kernel\|fused_f_kernel<1>\|fused_f_kernel<0>" $OUT/${p}_summary.txt | head -40; done
