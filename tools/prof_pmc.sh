#!/bin/bash
# Collect rocprofv3 PMC counters for one plan, one counter group per run (gpurun refuses --pmc with tracing).
# usage: tools/prof_pmc.sh <plan> <outdir> [command ...]      (default command: tools/run_encode.py --plan <plan> --steps 2)
set -u
PLAN=${1:-0}; OUT=${2:-gpurun_out/pmc}
shift; shift
REPO=$(pwd); export TMPDIR=/tmp
mkdir -p "$OUT"
if [ $# -gt 0 ]; then CMD=("$@"); else CMD=(python "$REPO/tools/run_encode.py" --plan "$PLAN" --steps 2); fi
run() { # name counters...
  local name=$1; shift
  ( cd /tmp && rocprofv3 --pmc "$@" -d "$REPO/$OUT/$name" -o pmc --output-format csv -- "${CMD[@]}" ) > "$OUT/$name.log" 2>&1
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM
run grbm GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "fastecc" not in k: continue
        k = k.replace("void ", "").replace("fastecc::", "").replace("(anonymous namespace)::", "").split("(")[0]
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}
json.dump(res, open(out + "/summary.json", "w"), indent=1)
for k, cs in res.items():
    print(k)
    for c, v in sorted(cs.items()): print("   %-24s %.4g" % (c, v))
PY
