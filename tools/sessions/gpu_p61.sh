#!/bin/bash
# GF((2^61-1)^2) session: parity tests of the field, then the configs[4] bench line and its rocprofv3 kernel stats.
# usage: tools/sessions/gpu_p61.sh <tag>
set -u
TAG=${1:-p61}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_p61.py tests/test_gpu_sharded.py -m gpu -x -q > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest.log"
for plan in ${PLANS:-0}; do
timeout 900 python bench.py --field p61 --steps 5 --warmup 1 --plan $plan ${BENCH_ARGS:-} > "$OUT/bench_p61_plan$plan.json" 2> "$OUT/bench_p61_plan$plan.err"; echo "bench plan $plan rc=$?"
python - "$OUT/bench_p61_plan$plan.json" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]
    print(d["value"],"GB/s",d["ms_per_step"],"ms",d["config"]["plan"]); print("  ",r["per_kernel_avg_ms"])
except Exception as e: print("no line:",e)
PY
done
timeout 900 bash tools/prof_stats.sh "$OUT/stats" python $(pwd)/bench.py --field p61 --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/stats.txt" 2>&1; grep -E "fastecc|Name" "$OUT/stats.txt" | head -12
