#!/bin/bash
# kernel trace of the HEADLINE mode (4 streams, one hipGraph replay per step): bash tools/trace_graph.sh <tag> [extra bench args]
TAG=${1:-g}; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-full-path --no-uvg --no-train-step "$@" > $OUT/trace.log 2>&1
tail -1 $OUT/trace.log > $OUT/bench_line_under_trace.json
python3 tools/trace_overlap.py $OUT/trace 0.0 > $OUT/graph_overlap_all.txt 2>&1
python3 tools/trace_overlap.py $OUT/trace 0.3 > $OUT/graph_overlap_steady.txt 2>&1
cat $OUT/graph_overlap_steady.txt
find $OUT -name "*.csv" -size +24M -delete
