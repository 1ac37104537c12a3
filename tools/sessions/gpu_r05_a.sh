#!/bin/bash
# Round 5, session A: the new N>1 bench control flow + ADVICE fixes on a GPU, SQ counters of the default plan's three kernels,
# direct-kernel counters per output count, the 8-rank control-flow run at the headline size.
set -u
TAG=${1:-r05a}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
R=$(pwd)
( rocminfo | grep -E "Marketing Name|Compute Unit|gfx" | head -8; nproc; cat /sys/fs/cgroup/cpu.max ) > "$OUT/box.txt" 2>&1
timeout 900 python -m pytest tests/test_gpu_bench_multirank.py tests/test_distributed.py tests/test_gpu_sharded.py tests/test_gpu_parity.py -m gpu -x -q > "$OUT/pytest_new.log" 2>&1; echo "pytest rc=$?"; tail -5 "$OUT/pytest_new.log"
# SQ counters, default plan, headline size (one counter group per run)
timeout 900 bash tools/prof_pmc.sh 0 "$OUT/pmc_default" > "$OUT/pmc_default.txt" 2>&1; tail -70 "$OUT/pmc_default.txt"
# kernel durations of the same command, un-instrumented by counters (for the effective clock: GRBM cycles of the PMC pass / its own duration is not available)
timeout 600 bash tools/prof_stats.sh "$OUT/stats" > "$OUT/stats.txt" 2>&1; grep -E "fastecc" "$OUT/stats.txt" | head -4 | cut -c1-220
timeout 900 python bench.py > "$OUT/bench_n1_default.json" 2> "$OUT/bench_n1_default.err"; echo "bench rc=$?"; cut -c1-600 "$OUT/bench_n1_default.json"
# direct MFMA kernel: counters per output count (separate processes, so the launches are not averaged together)
for e in 64 128; do
  for grp in "sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
             "sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM" \
             "sq3 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_I8" \
             "fetch FETCH_SIZE" "write WRITE_SIZE"; do
    set -- $grp; name=$1; shift
    ( cd /tmp && timeout 300 rocprofv3 --pmc "$@" -d "$R/$OUT/pmc_direct_$e/$name" -o pmc --output-format csv -- python "$R/tools/run_direct_mfma.py" $e ) > "$OUT/pmc_direct_${e}_$name.log" 2>&1
  done
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
res = {}
for e in (64, 128):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("%s/pmc_direct_%d/**/*counter_collection.csv" % (out, e), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")
            if "direct_" not in k and "sum_partials" not in k: continue
            k = k.replace("void ", "").replace("fastecc::", "").replace("(anonymous namespace)::", "").split("(")[0]
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    res["%d outputs" % e] = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}
json.dump(res, open(out + "/pmc_direct_by_outputs.json", "w"), indent=1)
print(json.dumps(res)[:3000])
PY
timeout 300 python tools/bench_direct.py 19 16,64,128 > "$OUT/direct_bench.jsonl" 2> "$OUT/direct.err"; tail -3 "$OUT/direct_bench.jsonl" | cut -c1-300
# configs[3] geometry: 8 ranks on this one GPU through gloo at the headline size (control flow + hash gate; times are host-staging times)
FASTECC_BENCH_BACKEND=gloo OMP_NUM_THREADS=2 timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29688 \
    bench.py --gpus 8 --steps 2 --warmup 1 --mode-timeout 400 --sharded-timeout 1000 > "$OUT/bench_8rank_gloo_one_gpu_control_flow.json" 2> "$OUT/bench_8rank.err"
echo "bench 8 ranks rc=$?"; cut -c1-400 "$OUT/bench_8rank_gloo_one_gpu_control_flow.json"; tail -3 "$OUT/bench_8rank.err"
