#!/bin/bash
# SQ / LDS counters of the GF((2^61-1)^2) tile kernels, one counter group per run (gpurun refuses --pmc with tracing).
# usage: tools/prof_pmc_p61.sh <outdir>
set -u
OUT=${1:-gpurun_out/pmc_p61}
REPO=$(pwd); export TMPDIR=/tmp
mkdir -p "$OUT"
run() { local name=$1; shift
  ( cd /tmp && rocprofv3 --pmc "$@" -d "$REPO/$OUT/$name" -o pmc --output-format csv -- python "$REPO/tools/run_encode.py" --field p61 --block-bytes 65536 --steps 1 ) > "$OUT/$name.log" 2>&1
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY
run sq2 SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM
run grbm GRBM_GUI_ACTIVE
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "p61_tile" not in k: continue
        k = k.replace("void ", "").replace("fastecc::", "").replace("p61::", "").replace("(anonymous namespace)::", "").split("(")[0]
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res))
PY
