#!/bin/bash
# Round 5, session B: direct MFMA kernel variants (XOR digits, one N-tile per wave at two waves per SIMD), A/B in separate processes.
# NOTE: ran against an EXPERIMENTAL build of csrc/direct.hip (XOR digits, one N-tile per wave: env FASTECC_DIRECT_NT / _G) that was measured null and is not in
# the tree; kept as the record of how profiles/r05/direct_mfma_variants_*.jsonl were made.
set -u
TAG=${1:-r05b}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_direct.py tests/test_gpu_decode.py -m gpu -x -q > "$OUT/pytest_direct.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest_direct.log"
for v in "NT=2 G=4" "NT=1 G=4"; do
  set -- $v; nt=${1#NT=}; g=${2#G=}
  echo "== $v"
  FASTECC_DIRECT_NT=$nt FASTECC_DIRECT_G=$g FASTECC_BENCH_DIRECT_FAST=1 timeout 300 python tools/bench_direct.py 19 48,64,96,128,256 2> "$OUT/direct_nt${nt}_g${g}.err" | tee "$OUT/direct_nt${nt}_g${g}.jsonl" | cut -c1-330
  FASTECC_DIRECT_NT=$nt FASTECC_DIRECT_G=$g timeout 600 python -m pytest tests/test_gpu_direct.py -m gpu -x -q 2>&1 | tail -1
done
