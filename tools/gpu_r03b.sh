#!/bin/bash
set -u
OUT=gpurun_out/r03b; mkdir -p $OUT; export TMPDIR=/tmp; R=$(pwd)
FASTECC_TRACE_PREPARE=1 timeout 300 python tools/trace_prepare.py > $OUT/trace_prepare.txt 2>&1; cat $OUT/trace_prepare.txt | grep -v amdgpu.ids
timeout 600 python bench.py --no-other-paths > $OUT/bench_default.json 2> $OUT/bench_default.err; python - $OUT/bench_default.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("bench", r["value"], r["ms_per_step"], r["roofline"]["per_kernel_avg_ms"], r["roofline"]["frac"], r["roofline"].get("frac_rocprof"), r["sharded_one_stripe"].get("with_gather"))
PY
timeout 600 bash tools/prof_stats.sh $OUT/stats python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-sharded --no-parity-check --no-other-paths > $OUT/stats.txt 2>&1; grep -E "ntt_tile" $OUT/stats.txt | cut -c1-160
for E in 16 64; do bash tools/prof_traffic.sh $OUT/t$E python $R/tools/run_direct_decode.py $E 2 > $OUT/traffic$E.json 2>&1; tail -1 $OUT/traffic$E.json | cut -c1-1500; done
bash tools/prof_stats.sh $OUT/sd64 python $R/tools/run_direct_decode.py 64 2 20 > $OUT/stats_direct64.txt 2>&1; grep -E "direct_|mfma_|interp" $OUT/stats_direct64.txt | cut -c1-200
