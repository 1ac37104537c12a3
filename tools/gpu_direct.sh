#!/bin/bash
# GPU-box session for direct.hip: its tests, then the timings.  usage: tools/gpu_direct.sh <tag>
set -u
TAG=${1:-direct}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_direct.py -x -q > "$OUT/pytest_direct.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/pytest_direct.log"
tail -15 "$OUT/pytest_direct.log"
timeout 600 python tools/bench_direct.py > "$OUT/bench_direct.jsonl" 2> "$OUT/bench_direct.err"; echo "bench rc=$?"; cat "$OUT/bench_direct.jsonl"; tail -3 "$OUT/bench_direct.err"
