#!/bin/bash
# GPU session: mixed-radix decoder tests, and the N = 2 control flow of bench.py (gloo hook on one GPU) with and without a stalled rank.
set -u
OUT=gpurun_out/${1:-r02t}
mkdir -p "$OUT"
timeout 1200 python -m pytest tests/test_gpu_mixed.py -x -q -m gpu > "$OUT/mixed.log" 2>&1; echo "pytest rc=$?"; tail -15 "$OUT/mixed.log"
export FASTECC_BENCH_BACKEND=gloo
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 > "$OUT/bench2.json" 2> "$OUT/bench2.err"; echo "bench2 rc=$?"; cut -c1-1500 "$OUT/bench2.json"; tail -3 "$OUT/bench2.err"
FASTECC_BENCH_TEST_STALL=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 5 --warmup 2 --sharded-timeout 25 > "$OUT/bench2_stall.json" 2> "$OUT/bench2_stall.err"; echo "bench2 stall rc=$?"; python -c "
import json,sys
d=json.loads(open('$OUT/bench2_stall.json').read().strip().splitlines()[-1]); print(d['value'], d['sharded_one_stripe'], d['parity_check'])"; tail -3 "$OUT/bench2_stall.err"
