#!/bin/bash
# Round 4, closing session: the evidence DESIGN.md cites, all from ONE box and HEAD.
set -u
TAG=${1:-r04final}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
R=$(pwd)
( rocminfo | grep -E "Marketing Name|Compute Unit|gfx" | head -8; rocm-smi --showmeminfo vram | head -8; nproc; cat /sys/fs/cgroup/cpu.max ) > "$OUT/box.txt" 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/pytest_gpu.log"; tail -3 "$OUT/pytest_gpu.log"
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -1 "$OUT/smoke.log"
timeout 900 python bench.py > "$OUT/bench_n1_default.json" 2> "$OUT/bench_n1_default.err"; echo "bench rc=$?"; cut -c1-700 "$OUT/bench_n1_default.json"
timeout 600 bash tools/prof_stats.sh "$OUT/stats" > "$OUT/stats.txt" 2>&1; grep -E "fastecc" "$OUT/stats.txt" | head -4 | cut -c1-200
timeout 900 bash tools/prof_traffic.sh "$OUT/traffic_default" python $R/tools/run_encode.py --steps 2 > "$OUT/traffic_default.txt" 2>&1; tail -1 "$OUT/traffic_default.txt" | cut -c1-600
for n in 2 4; do
  FASTECC_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29610 + n)) \
      bench.py --gpus $n --steps 3 --warmup 1 > "$OUT/bench_${n}rank_gloo_one_gpu_control_flow.json" 2> "$OUT/bench_${n}rank.err"
  echo "bench $n ranks rc=$?"
done
timeout 300 fastecc_amd/lib/rs_hip 19 4096 > "$OUT/rs_hip.log" 2>&1; tail -2 "$OUT/rs_hip.log"
timeout 300 fastecc_amd/lib/rs_hip 19 4096 gpus=0,0,0,0,0,0,0,0 > "$OUT/rs_hip_sharded.log" 2>&1; tail -2 "$OUT/rs_hip_sharded.log"
timeout 300 python tools/bench_host_link.py > "$OUT/host_link_and_pipeline.jsonl" 2> "$OUT/host_link.err"; cat "$OUT/host_link_and_pipeline.jsonl" | cut -c1-400
timeout 600 python tools/bench_direct.py 19 1,16,32,64,128,256 > "$OUT/direct_bench.jsonl" 2> "$OUT/direct.err"; tail -2 "$OUT/direct_bench.jsonl" | cut -c1-300
for n in 19 18 17 16; do timeout 300 python tools/sweep_plans.py --log2k $n --plans 0,0,3100,3090,3080 --steps 20 2>/dev/null | grep "^{" | sed "s/^{/{\"log2k\": $n, /"; done > "$OUT/plan_sweep_mid_levels.jsonl"; cut -c1-200 "$OUT/plan_sweep_mid_levels.jsonl" | head -6
timeout 300 python tools/bench_decode_forms.py 2>/dev/null | grep "^{" > "$OUT/decode_forms.jsonl"; cut -c1-150 "$OUT/decode_forms.jsonl"
timeout 600 python tools/bench_mixed.py > "$OUT/mixed_radix_bench.jsonl" 2>/dev/null; timeout 600 python tools/bench_mixed_pfa.py > "$OUT/mixed_pfa_bench.jsonl" 2>/dev/null; wc -l "$OUT"/mixed_*.jsonl
timeout 300 python tools/bench_blocks.py 2>/dev/null | grep "^{" > "$OUT/encode_blocks_now.jsonl"; cut -c1-200 "$OUT/encode_blocks_now.jsonl"
bash tools/trace_host_pinned.sh "$OUT/trace_host" > "$OUT/host_pinned_copy_trace_after.txt" 2>&1; head -12 "$OUT/host_pinned_copy_trace_after.txt"
