#!/bin/bash
# Round 6, closing session: the evidence DESIGN.md cites, from ONE box and the tree as it is.
set -u
TAG=${1:-r06final}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
R=$(pwd)
( rocminfo | grep -E "Marketing Name|Compute Unit|gfx" | head -8; rocm-smi --showmeminfo vram | head -8; nproc; cat /sys/fs/cgroup/cpu.max ) > "$OUT/box.txt" 2>&1
# 1. the counter files bench.py quotes, stamped with the sources they describe
bash tools/sessions/gpu_r06_counters.sh "$TAG/counters" > "$OUT/counters.log" 2>&1; tail -3 "$OUT/counters.log" | cut -c1-200
mkdir -p "$OUT/profiles"; cp "$OUT/counters/stamped/"* "$OUT/profiles/" 2>/dev/null
# (bench.py below must see them: on the box they go where the tree keeps them)
cp "$OUT/counters/stamped/"* profiles/r06/ 2>/dev/null
# 2. tests, smoke, the driver's line
timeout 2400 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/pytest_gpu.log"; tail -3 "$OUT/pytest_gpu.log"
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -1 "$OUT/smoke.log"
( tail -3 "$OUT/pytest_gpu.log"; tail -1 "$OUT/smoke.log" ) > "$OUT/profiles/pytest_gpu_summary.txt"
timeout 900 python bench.py > "$OUT/bench_n1_default.json" 2> "$OUT/bench_n1_default.err"; echo "bench rc=$?"; cut -c1-400 "$OUT/bench_n1_default.json"
cp "$OUT/bench_n1_default.json" "$OUT/box.txt" "$OUT/profiles/"
# 3. the decoder's set-up, traced (cold process; after an encode as in bench.py)
bash tools/sessions/gpu_r06_prepare_trace.sh > "$OUT/prepare_cold.txt" 2>&1
( echo "== FASTECC_TRACE_PREPARE=1 python tools/run_prepare.py 0.02 6 (cold process) =="; grep -v amdgpu.ids gpurun_out/r06prep/trace_002.txt; echo; echo "== kernels of the last (steady) call: rocprofv3 --kernel-trace =="; cat "$OUT/prepare_cold.txt" ) > "$OUT/profiles/prepare_trace.txt"
bash tools/sessions/gpu_r06_prepare_trace_warm.sh > "$OUT/prepare_warm.txt" 2>&1
( echo "== first fastecc_decode_prepare after an encode (tools/run_prepare.py 0.02 3 warm): gaps before kernels, HIP API totals, slowest calls, phases =="; grep -v amdgpu.ids "$OUT/prepare_warm.txt" ) > "$OUT/profiles/prepare_trace_warm.txt"
bash tools/sessions/gpu_r06_prepare_trace_p61.sh > /dev/null 2>&1; cp gpurun_out/r06prep61/prepare_trace_p61.txt "$OUT/profiles/" 2>/dev/null
# 4. the 64-bit field's coset codes (n = 4k from bench.py's other_paths above; n = 8k here)
timeout 600 python tools/bench_p61_n8k.py 2> /dev/null | tail -1 > "$OUT/p61_n8k.json"; cut -c1-300 "$OUT/p61_n8k.json"
ls "$OUT/profiles"
