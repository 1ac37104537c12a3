#!/bin/bash
# Profiling session for profiles/<round>/: rocprofv3 --kernel-trace --stats of the two bench commands, and HBM traffic
# (FETCH_SIZE / WRITE_SIZE, one counter per run) of the kernels of both fields.
# usage: tools/sessions/gpu_profile.sh <tag>
set -u
TAG=${1:-prof}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
R=$(pwd)
timeout 600 bash tools/prof_stats.sh "$OUT/stats_default" > "$OUT/stats_default.txt" 2>&1; grep -E "fastecc" "$OUT/stats_default.txt" | head -5 | cut -c1-200
timeout 900 bash tools/prof_stats.sh "$OUT/stats_p61" python $R/bench.py --field p61 --steps 3 --warmup 1 --no-cpu-baseline --no-sharded > "$OUT/stats_p61.txt" 2>&1; grep -E "fastecc" "$OUT/stats_p61.txt" | head -6 | cut -c1-200
timeout 900 bash tools/prof_traffic.sh "$OUT/traffic_default" python $R/tools/run_encode.py --steps 2 > "$OUT/traffic_default.txt" 2>&1; tail -1 "$OUT/traffic_default.txt" | cut -c1-600
timeout 900 bash tools/prof_traffic.sh "$OUT/traffic_p61" python $R/tools/run_encode.py --field p61 --block-bytes 65536 --steps 1 > "$OUT/traffic_p61.txt" 2>&1; tail -1 "$OUT/traffic_p61.txt" | cut -c1-900
