#!/bin/bash
# HBM traffic per kernel launch from rocprofv3 PMC counters, one counter per run (gpurun refuses --pmc with tracing).
# usage: tools/prof_traffic.sh <outdir> <command ...>
# Prints / writes <outdir>/traffic.json: kernel -> bytes per launch = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024
# (KB units; FETCH_SIZE counts 64 B per 128 B request on gfx950, MI355X_MICROARCH.md HBM section).
set -u
OUT=$1; shift
REPO=$(pwd); export TMPDIR=/tmp
mkdir -p "$OUT"
for ctr in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rocprofv3 --pmc $ctr -d "$REPO/$OUT/$ctr" -o pmc --output-format csv -- "$@" ) > "$OUT/$ctr.log" 2>&1
done
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
res = {}
for k, cs in agg.items():
    f = sum(cs.get("FETCH_SIZE", [0])) / max(1, len(cs.get("FETCH_SIZE", [0])))
    w = sum(cs.get("WRITE_SIZE", [0])) / max(1, len(cs.get("WRITE_SIZE", [0])))
    res[k] = {"hbm_bytes_per_launch": int(2 * f * 1024 + w * 1024), "fetch_KB_raw": f, "write_KB": w, "launches": len(cs.get("FETCH_SIZE", []))}
json.dump(res, open(out + "/traffic.json", "w"), indent=1)
print(json.dumps(res))
PY
