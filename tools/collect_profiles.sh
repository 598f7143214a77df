#!/bin/bash
# After tools/experiments/r6_finalA.sh ran on a GPU box (tools/profile_gpu.sh 1 and 2 streams, tools/pmc_train.sh 8, replayed-step traces,
# tools/profile_uvg.sh): copy the summaries into profiles/<round>/ and derive the JSONs bench.py reads.   bash tools/collect_profiles.sh r6
set -eu
R=${1:-r6}; P=profiles/$R
mkdir -p $P
for t in 1 2; do d=gpurun_out/prof_${R}_${t}s; for f in config.txt kernel_trace_summary.txt pmc1_summary.txt pmc2_summary.txt pmc3_summary.txt pmc4_summary.txt pmc5_summary.txt; do cp $d/$f $P/${R}_${t}stream_$f; done; done
python3 tools/pmc_traffic.py gpurun_out/prof_${R}_2s $P/pmc_traffic.json "$P/${R}_2stream_pmc*_summary.txt (tools/profile_gpu.sh ${R}_2s 2)" 14 > /dev/null
python3 tools/rocprof_avgs.py gpurun_out/prof_${R}_1s/kernel_trace_summary.txt $P "tools/profile_gpu.sh ${R}_1s 1: rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-graph --streams 1 ..." > /dev/null
python3 tools/pmc_train.py gpurun_out/pmc_train $P > /dev/null
for f in kernel_trace_summary pmc1_summary pmc2_summary pmc3_summary pmc4_summary; do cp gpurun_out/pmc_train/$f.txt $P/train_step_$f.txt; done
cp gpurun_out/${R}trace_final/steps_b1.txt $P/train_step_replayed_b1.txt; cp gpurun_out/${R}trace_final/steps_b8.txt $P/train_step_replayed_b8.txt
d=gpurun_out/prof_uvg_$R; cp $d/kernel_trace_1080p.txt $P/uvg1080p_1stream_kernel_trace.txt; cp $d/kernel_trace_1080p_2streams.txt $P/uvg1080p_2streams_kernel_trace.txt; cp $d/kernel_trace_256x448_one_gop.txt $P/fullpath_256x448_one_gop_kernel_trace.txt
for k in fetch write tcc sq; do cp $d/pmc_${k}_summary.txt $P/uvg1080p_pmc_$k.txt; done
python3 tools/pmc_uvg.py $P > /dev/null
python3 - <<PY
import json
for f in ("pmc_traffic.json", "rocprof_kernel_avgs.json", "train_step_pmc.json", "uvg1080p_pmc.json"):
    print(f, json.load(open("$P/" + f))["_meta"].get("csrc_sha16"))
PY
