#!/bin/bash
# One GPU-box session: full -m gpu suite, smoke, the default bench line, rocprofv3 --kernel-trace --stats of the same
# command, and the C++ host driver.  Everything lands in gpurun_out/<tag>/.
# usage: tools/sessions/gpu_round.sh <tag> [pytest args...]
set -u
TAG=${1:-run}; shift || true
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
( rocminfo | grep -E "Marketing Name|Compute Unit|gfx" | head -8; rocm-smi --showmeminfo vram | head -8; nproc ) > "$OUT/box.txt" 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q "$@" > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/pytest_gpu.log"
tail -5 "$OUT/pytest_gpu.log"
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -2 "$OUT/smoke.log"
timeout 600 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench rc=$?"; cat "$OUT/bench_default.json"
timeout 600 bash tools/prof_stats.sh "$OUT/stats" > "$OUT/stats.txt" 2>&1; grep -E "fastecc|Name" "$OUT/stats.txt" | head -12
timeout 300 fastecc_amd/lib/rs_hip 19 4096 > "$OUT/rs_hip.log" 2>&1; tail -3 "$OUT/rs_hip.log"
timeout 300 fastecc_amd/lib/rs_hip 19 4096 gpus=0,0,0,0,0,0,0,0 > "$OUT/rs_hip_sharded.log" 2>&1; tail -4 "$OUT/rs_hip_sharded.log"
[ -x oracle/_ref/rs-hip-patched ] && { timeout 300 oracle/_ref/rs-hip-patched 19 4096 > "$OUT/rs_hip_patched.log" 2>&1; tail -2 "$OUT/rs_hip_patched.log"; }
