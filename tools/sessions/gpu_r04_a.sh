#!/bin/bash
# Round 4, session A: the full GPU suite, the default bench line + its rocprofv3 summary, the multi-rank control flow on one GPU
# (gloo, every rank on device 0), the column-slab overlap experiment on the headline, HBM traffic of the default plan.
# usage: tools/sessions/gpu_r04_a.sh <tag>
set -u
TAG=${1:-r04a}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
R=$(pwd)
( rocminfo | grep -E "Marketing Name|Compute Unit|gfx" | head -8; rocm-smi --showmeminfo vram | head -8; nproc; cat /sys/fs/cgroup/cpu.max ) > "$OUT/box.txt" 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/pytest_gpu.log"; tail -4 "$OUT/pytest_gpu.log"
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -2 "$OUT/smoke.log"
timeout 900 python bench.py > "$OUT/bench_n1_default.json" 2> "$OUT/bench_n1_default.err"; echo "bench rc=$?"; cut -c1-1500 "$OUT/bench_n1_default.json"
timeout 600 bash tools/prof_stats.sh "$OUT/stats" > "$OUT/stats.txt" 2>&1; grep -E "fastecc|Name" "$OUT/stats.txt" | head -8 | cut -c1-220
# the N > 1 line's control flow on one GPU: 2 and 4 ranks, gloo through host memory, the headline stripe (reference hash gate inside)
for n in 2 4; do
  FASTECC_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29510 + n)) \
      bench.py --gpus $n --steps 3 --warmup 1 > "$OUT/bench_${n}rank_gloo_one_gpu_control_flow.json" 2> "$OUT/bench_${n}rank.err"
  echo "bench $n ranks rc=$?"; tail -1 "$OUT/bench_${n}rank_gloo_one_gpu_control_flow.json" | cut -c1-2500
done
# headline overlap experiment: H column slabs on internal streams, each one pass behind the previous (MID of one slab beside the outer passes of the next)
for h in 0 2 4 8; do
  timeout 300 python bench.py --steps 30 --warmup 5 --slabs $h --no-cpu-baseline --no-sharded --no-other-paths > "$OUT/bench_slabs$h.json" 2>> "$OUT/bench_slabs.err"
  python - "$OUT/bench_slabs$h.json" $h <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("slabs", sys.argv[2], "ms_per_step", d["ms_per_step"], "value", d["value"], "parity", (d.get("parity_check") or {}).get("status"), d["roofline"]["per_kernel_avg_ms"])
PY
done
( cd /tmp && rocprofv3 --kernel-trace -d "$R/$OUT/trace_slabs2" -o t --output-format csv -- python "$R/bench.py" --steps 4 --warmup 2 --slabs 2 --no-cpu-baseline --no-sharded --no-other-paths --no-parity-check ) > "$OUT/trace_slabs2.log" 2>&1
f=$(find "$OUT/trace_slabs2" -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python - "$f" > "$OUT/trace_slabs2_overlap.txt" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "ntt_tile_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-12:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    n = r["Kernel_Name"]; n = n[n.index("<"):n.index(">") + 1]
    print("%-28s stream %-4s start %9.1f us  end %9.1f us" % (n, r.get("Stream_Id", r.get("Queue_Id", "?")), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3))
PY
cat "$OUT/trace_slabs2_overlap.txt" 2>/dev/null | head -14
bash tools/power_probe.sh "$OUT/power" > "$OUT/power.txt" 2>&1; tail -12 "$OUT/power.txt"
timeout 900 bash tools/prof_traffic.sh "$OUT/traffic_default" python $R/tools/run_encode.py --steps 2 > "$OUT/traffic_default.txt" 2>&1; tail -1 "$OUT/traffic_default.txt" | cut -c1-600
