#!/bin/bash
# GPU-box extras of round 3: radix-4 microbenchmark, the p61 bench line, a 2-rank gloo run of bench.py's multi-rank control flow (both ranks
# on device 0), the direct-path timings.  usage: tools/sessions/gpu_extras.sh <tag>
set -u
TAG=${1:-extras}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 120 fastecc_amd/lib/microbench radix > "$OUT/microbench_radix.jsonl" 2>&1; cat "$OUT/microbench_radix.jsonl"
timeout 600 python bench.py --field p61 --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/bench_p61.json" 2> "$OUT/bench_p61.err"; echo "p61 rc=$?"; python - "$OUT/bench_p61.json" <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("p61:", r["value"], "GB/s", r["ms_per_step"], "ms", r["roofline"]["per_kernel_avg_ms"], r["parity_check"]["status"])
except Exception as e:
    print("p61 parse failed", e)
PY
FASTECC_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --log2k 16 > "$OUT/bench_gloo2.json" 2> "$OUT/bench_gloo2.err"; echo "gloo2 rc=$?"; tail -c 1800 "$OUT/bench_gloo2.json"; tail -3 "$OUT/bench_gloo2.err"
FASTECC_TRACE_PREPARE=1 timeout 300 python tools/trace_prepare.py > "$OUT/prepare_trace.txt" 2>&1; grep -v amdgpu "$OUT/prepare_trace.txt"
R=$(pwd)
for E in 16 64 256; do bash tools/prof_traffic.sh "$OUT/t$E" python $R/tools/run_direct_decode.py $E 2 > "$OUT/traffic$E.json" 2>&1; done
timeout 300 python tools/bench_decode.py > "$OUT/decode_bench.json" 2> "$OUT/decode_bench.err"; tail -c 600 "$OUT/decode_bench.json"; echo
timeout 600 python tools/bench_direct.py > "$OUT/direct_bench.jsonl" 2> "$OUT/direct_bench.err"; echo "direct rc=$?"; python - "$OUT/direct_bench.jsonl" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    r = json.loads(l)
    if r["case"] == "decode":
        print("decode", r["lost_data_blocks"], {k: (v["decode_ms"], v["prepare_ms"], v["ok"]) for k, v in r.items() if isinstance(v, dict)})
    else:
        print("encode", r["parity_blocks"], {k: (v["ms"], v["same_as_pipeline"]) for k, v in r.items() if isinstance(v, dict)})
PY
