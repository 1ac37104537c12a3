#!/bin/bash
# rocprofv3 --kernel-trace --stats of a bench run; summary copied by the caller into profiles/.
# usage: tools/prof_stats.sh <outdir> [command ...]      (default command: the default bench.py run)
set -u
OUT=${1:-gpurun_out/stats}
shift || true
REPO=$(pwd); export TMPDIR=/tmp
if [ $# -eq 0 ]; then set -- python "$REPO/bench.py" --steps 50 --warmup 5 --no-cpu-baseline --no-sharded --no-parity-check --no-other-paths; fi
mkdir -p "$OUT"
( cd /tmp && rocprofv3 --kernel-trace --stats -d "$REPO/$OUT" -o bench --output-format csv -- "$@" ) > "$OUT/run.log" 2>&1
ls "$OUT" >> "$OUT/run.log"
f=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv" && cat "$OUT/kernel_stats.csv"
