#!/bin/bash
# Round 5: counters of the 64-bit field's kernels at the configs[4] size, and of its isolated butterfly loops (the VALU issue yardstick)
set -u
OUT=gpurun_out/r05p61; mkdir -p $OUT; export TMPDIR=/tmp; R=$(pwd)
bash tools/prof_pmc_p61.sh $OUT/pmc > $OUT/pmc.txt 2>&1
( cd /tmp && rocprofv3 --kernel-trace --stats -d $R/$OUT/stats -o st --output-format csv -- python $R/tools/run_encode.py --field p61 --block-bytes 65536 --steps 3 ) > $OUT/stats.log 2>&1
( cd /tmp && rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES -d $R/$OUT/pmc_loop -o pmc --output-format csv -- $R/fastecc_amd/lib/microbench_p61 ) > $OUT/pmc_loop.log 2>&1
fastecc_amd/lib/microbench_p61 > $OUT/microbench_p61.jsonl 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, json, collections
out = sys.argv[1]
pm = json.load(open(out + "/pmc/summary.json"))
dur = {}
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "p61_tile" in r["Name"]:
            dur[r["Name"].replace("void ", "").replace("fastecc::", "").replace("p61::", "").replace("(anonymous namespace)::", "").split("(")[0]] = float(r["AverageNs"]) / 1e6
loops = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc_loop/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        loops[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {"kernels": {}, "isolated_loops": {}}
for k, c in loops.items():
    if "SQ_INSTS_VALU" in c and c["SQ_INSTS_VALU"][0] > 1e6:
        cyc = sum(c["GRBM_GUI_ACTIVE"]) / len(c["GRBM_GUI_ACTIVE"]) / 8
        res["isolated_loops"][k] = {"valu_per_simd_cycle": c["SQ_INSTS_VALU"][0] / (1024 * cyc)}
for k, c in pm.items():
    cyc = c["GRBM_GUI_ACTIVE"] / 8
    e = {"cycles": cyc, "valu_per_simd_cycle": c["SQ_INSTS_VALU"] / (1024 * cyc), "raw": c}
    if k in dur:
        e["duration_ms"] = dur[k]; e["clock_GHz"] = cyc / dur[k] / 1e6
        e["hbm_GBps_algorithmic"] = 2 * (1 << 19) * 65536 / dur[k] / 1e6
    res["kernels"][k] = e
json.dump(res, open(out + "/pmc_p61_summary.json", "w"), indent=1)
for k, e in res["kernels"].items(): print(k, {x: round(v, 4) for x, v in e.items() if x != "raw"})
for k, e in res["isolated_loops"].items(): print(k, e)
PY
grep p61_bfly $OUT/microbench_p61.jsonl | cut -c1-200
