#!/bin/bash
# Round 5, session C: what the matrix cores and the VALU issue at most — the yardsticks behind roofline.bound
set -u
TAG=${1:-r05c}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; R=$(pwd)
fastecc_amd/lib/microbench_mfma > "$OUT/microbench_mfma.jsonl" 2>&1; cat "$OUT/microbench_mfma.jsonl" | cut -c1-330
( cd /tmp && rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d "$R/$OUT/pmc_mfma" -o pmc --output-format csv -- "$R/fastecc_amd/lib/microbench_mfma" ) > "$OUT/pmc_mfma.log" 2>&1
( cd /tmp && rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d "$R/$OUT/pmc_bfly" -o pmc --output-format csv -- "$R/fastecc_amd/lib/microbench" bfly radix ) > "$OUT/pmc_bfly.log" 2>&1
fastecc_amd/lib/microbench bfly radix > "$OUT/microbench_bfly.jsonl" 2>&1; grep -E "radix|mad64" "$OUT/microbench_bfly.jsonl" | cut -c1-200 | head -8
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
for sub in ("pmc_mfma", "pmc_bfly"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (out, sub), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "").split("(")[0].replace("void ", "")
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    res = {k: dict({c: v for c, v in cs.items()}) for k, cs in agg.items()}
    json.dump(res, open("%s/%s_by_dispatch.json" % (out, sub), "w"), indent=1)
    for k, cs in res.items():
        print(k[:90])
        for c, v in sorted(cs.items()): print("   %-28s %s" % (c, ["%.4g" % x for x in v][:10]))
PY
