#!/bin/bash
# Runs on the GPU box: kernel trace of the STP chain alone (tools/trace_stp.py), 4 clips on one stream and 1 clip (what one
# stream of the 4-stream pipeline runs).  Output: gpurun_out/stp{4,1}_summary.txt
set -u
export TMPDIR=/tmp
for c in 4 1; do
  rm -rf gpurun_out/stp$c; mkdir -p gpurun_out/stp$c
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stp$c -- python3 tools/trace_stp.py $c > gpurun_out/stp$c.log 2>&1 || exit 1
  python3 tools/prof_summary.py gpurun_out/stp$c > gpurun_out/stp${c}_summary.txt
  find gpurun_out/stp$c -name "*.csv" -size +1M -delete
  grep "stp chain" gpurun_out/stp$c.log
  grep -v "at::native\|rocclr\|Cijk" gpurun_out/stp${c}_summary.txt | cut -c1-130 | head -12
done
