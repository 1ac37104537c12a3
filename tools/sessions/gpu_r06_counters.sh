#!/bin/bash
# Round 6: the counter files bench.py quotes, all from ONE box and the tree as it is, stamped with the sources they describe.
#   profiles/r06/{rocprofv3_kernel_stats_bench_default.csv, pmc_default_plan_raw.json, pmc_microbench_bfly.json, pmc_valu_default_plan.json,
#                 pmc_traffic.json, counters_stamp.json}
set -u
TAG=${1:-r06counters}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; R=$(pwd)
timeout 600 bash tools/prof_stats.sh "$OUT/stats" > "$OUT/stats.txt" 2>&1; grep -E "ntt_tile" "$OUT/stats.txt" | head -4 | cut -c1-160
timeout 900 bash tools/prof_pmc.sh 0 "$OUT/pmc_default" > "$OUT/pmc_default.txt" 2>&1; tail -5 "$OUT/pmc_default.txt"
( cd /tmp && rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d "$R/$OUT/pmc_bfly" -o pmc --output-format csv -- "$R/fastecc_amd/lib/microbench" bfly radix ) > "$OUT/pmc_bfly.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json, re
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("%s/pmc_bfly/**/*counter_collection.csv" % out, recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "").split("(")[0].replace("void ", "")
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
json.dump({k: dict(cs) for k, cs in agg.items()}, open("%s/pmc_microbench_bfly.json" % out, "w"), indent=1)
# traffic per launch, by the library's profile names
summ = json.load(open("%s/pmc_default/summary.json" % out))
def pname(t):
    m = re.search(r"ntt_tile_kernel<(\d+), (\d+), (true|false), (\d+)", t)
    if not m: return None
    return "tile_%s%s_w%s%s" % ({"0": "dif", "1": "dit", "2": "mid"}[m.group(4)], m.group(1), "32" if m.group(3) == "true" else "64", "" if m.group(2) == "5" else "_r16")
tr = {"_method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs (tools/prof_pmc.sh 0, session tools/sessions/gpu_r06_counters.sh), KB units; HBM bytes per launch = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950: FETCH_SIZE counts 64 B per 128 B request, MI355X_MICROARCH.md HBM section); algorithmic bytes per launch 4294967296"}
for k, c in summ.items():
    n = pname(k)
    if n and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        tr[n] = int(2 * c["FETCH_SIZE"] * 1024 + c["WRITE_SIZE"] * 1024)
        tr["_raw_" + n] = {"FETCH_SIZE_KB": c["FETCH_SIZE"], "WRITE_SIZE_KB": c["WRITE_SIZE"]}
json.dump(tr, open("%s/pmc_traffic.json" % out, "w"), indent=1)
print({k: v for k, v in tr.items() if not k.startswith("_")})
PY
cp "$OUT/stats/kernel_stats.csv" "$OUT/rocprofv3_kernel_stats_bench_default.csv"
cp "$OUT/pmc_default/summary.json" "$OUT/pmc_default_plan_raw.json"
python3 tools/pmc_valu.py "$OUT/pmc_default_plan_raw.json" "$OUT/rocprofv3_kernel_stats_bench_default.csv" "$OUT/pmc_microbench_bfly.json" "$OUT/pmc_valu_default_plan.json" 2>&1 | tail -5
mkdir -p "$OUT/stamped"; cp "$OUT/rocprofv3_kernel_stats_bench_default.csv" "$OUT/pmc_default_plan_raw.json" "$OUT/pmc_microbench_bfly.json" "$OUT/pmc_valu_default_plan.json" "$OUT/pmc_traffic.json" "$OUT/stamped/"
python3 tools/stamp_counters.py "$OUT/stamped" "tools/sessions/gpu_r06_counters.sh" | cut -c1-300
