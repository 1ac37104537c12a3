#!/bin/bash
# Round 5, closing session: the evidence DESIGN.md cites, from ONE box and HEAD.
set -u
TAG=${1:-r05final}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
R=$(pwd)
( rocminfo | grep -E "Marketing Name|Compute Unit|gfx" | head -8; rocm-smi --showmeminfo vram | head -8; nproc; cat /sys/fs/cgroup/cpu.max ) > "$OUT/box.txt" 2>&1
timeout 1800 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/pytest_gpu.log"; tail -3 "$OUT/pytest_gpu.log"
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -1 "$OUT/smoke.log"
timeout 900 python bench.py > "$OUT/bench_n1_default.json" 2> "$OUT/bench_n1_default.err"; echo "bench rc=$?"; cut -c1-500 "$OUT/bench_n1_default.json"
timeout 600 bash tools/prof_stats.sh "$OUT/stats" > "$OUT/stats.txt" 2>&1; grep -E "fastecc" "$OUT/stats.txt" | head -4 | cut -c1-200
