#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default bench.py run; summary copied by the caller into profiles/.
# usage: tools/prof_stats.sh <outdir>
set -u
OUT=${1:-gpurun_out/stats}
REPO=$(pwd); export TMPDIR=/tmp
mkdir -p "$OUT"
( cd /tmp && rocprofv3 --kernel-trace --stats -d "$REPO/$OUT" -o bench --output-format csv -- python "$REPO/bench.py" --steps 10 --warmup 2 --no-cpu-baseline ) > "$OUT/run.log" 2>&1
ls "$OUT" >> "$OUT/run.log"
f=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv" && cat "$OUT/kernel_stats.csv"
