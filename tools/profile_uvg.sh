#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + traffic / LDS / issue counters of ONE 1080p GOP (config 5) through the whole test
# path, eager on one stream (tools/trace_uvg.py), and the same trace at the headline's size for a per-pixel-frame comparison.
# Output under gpurun_out/prof_uvg_<tag>/ ; summaries are copied into profiles/ by hand.  python3 sits directly behind `--`.
set -u
TAG=${1:-r5}
OUT=gpurun_out/prof_uvg_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
U="tools/trace_uvg.py --path full --streams 1"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace1080 -- python3 $U --reps 3 > $OUT/trace1080.log 2>&1
python3 tools/prof_summary.py $OUT/trace1080 > $OUT/kernel_trace_1080p.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace1080s2 -- python3 $U --streams 2 --reps 3 > $OUT/trace1080s2.log 2>&1
python3 tools/prof_summary.py $OUT/trace1080s2 > $OUT/kernel_trace_1080p_2streams.txt 2>&1
# the same pipeline at the headline's frame size, 4 GOPs (28 frames) on one stream: per-pixel-frame kernel times to compare with
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace256 -- python3 tools/trace_uvg.py --path full --streams 1 --height 256 --width 448 --reps 12 > $OUT/trace256.log 2>&1
python3 tools/prof_summary.py $OUT/trace256 > $OUT/kernel_trace_256x448_one_gop.txt 2>&1
P="$U --reps 1 --warm 1"
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $P > $OUT/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $P > $OUT/pmc_write.log 2>&1
timeout 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_tcc -- python3 $P > $OUT/pmc_tcc.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 $P > $OUT/pmc_sq.log 2>&1
for p in pmc_fetch pmc_write pmc_tcc pmc_sq; do python3 tools/prof_summary.py $OUT/$p --pmc > $OUT/${p}_summary.txt 2>&1; done
cat $OUT/trace1080.log | tail -1; cat $OUT/trace1080s2.log | tail -1; cat $OUT/trace256.log | tail -1
head -30 $OUT/kernel_trace_1080p.txt
find $OUT -name "*.csv" -size +1M -delete
