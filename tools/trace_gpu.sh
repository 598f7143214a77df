#!/bin/bash
# kernel trace only (quick): bash tools/trace_gpu.sh <tag> [extra bench args]
TAG=${1:-t}; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 5 --warmup 2 --no-graph --no-cpu-baseline --no-full-path --no-uvg --no-train-step --streams 1 "$@" > $OUT/trace.log 2>&1
python3 tools/prof_summary.py $OUT/trace > $OUT/kernel_trace_summary.txt 2>&1
cat $OUT/kernel_trace_summary.txt
find $OUT -name "*.csv" -size +1M -delete
