export TMPDIR=/tmp
OUT=gpurun_out/pmc_quick
rm -rf $OUT; mkdir -p $OUT
ARGS="bench.py --steps 1 --warmup 0 --no-graph --no-cpu-baseline --no-full-path --no-uvg --streams 1"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $OUT/p1 -- python3 $ARGS > $OUT/p1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/p2 -- python3 $ARGS > $OUT/p2.log 2>&1
python3 tools/prof_summary.py $OUT/p1 --pmc | grep -A9 -E "fused_gh|conv3x3_kernel<16, 16, 4, 2, 0>"
python3 tools/prof_summary.py $OUT/p2 --pmc | grep -A7 -E "fused_gh|conv3x3_kernel<16, 16, 4, 2, 0>"
find $OUT -name "*.csv" -size +1M -delete
