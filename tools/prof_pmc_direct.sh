#!/bin/bash
# SQ counters of direct_mfma_kernel (k + 64 / k + 128 encodes): where a single wave per SIMD spends its cycles
set -u
OUT=${1:-gpurun_out/pmc_direct}; mkdir -p "$OUT"; export TMPDIR=/tmp; R=$(pwd)
run() { local name=$1; shift; ( cd /tmp && rocprofv3 --pmc "$@" -d "$R/$OUT/$name" -o pmc --output-format csv -- python "$R/tools/run_direct_mfma.py" 64,128 ) > "$OUT/$name.log" 2>&1; }
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
run sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM
run sq3 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_I8 GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "direct_mfma" not in k and "direct_accumulate" not in k: continue
        k = k.replace("void ", "").replace("fastecc::", "").replace("(anonymous namespace)::", "").split("(")[0]
        agg[k + " grid=" + row.get("Grid_Size", "?")][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}
json.dump(res, open(out + "/summary.json", "w"), indent=1)
for k, cs in res.items():
    print(k)
    for c, v in sorted(cs.items()): print("   %-32s %.5g" % (c, v))
PY
