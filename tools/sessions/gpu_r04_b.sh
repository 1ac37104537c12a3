#!/bin/bash
# Round 4, session B: the reshaped fused odd-radix kernels — parity tests, then tools/bench_mixed.py (compare with profiles/r03/mixed_radix_bench.jsonl)
set -u
TAG=${1:-r04b}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_mixed.py tests/test_gpu_fuzz.py -x -q > "$OUT/pytest_mixed.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest_mixed.log"
timeout 600 python tools/bench_mixed.py > "$OUT/mixed_radix_bench.jsonl" 2> "$OUT/mixed.err"; echo "bench rc=$?"
python - "$OUT/mixed_radix_bench.jsonl" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    r = json.loads(l)
    mr, ze = r.get("mixed_radix", {}), r.get("zero_extended_pow2", {})
    print(r["q"], r["m"], mr.get("ms"), ze.get("ms"), mr.get("kernel_ms"))
PY
