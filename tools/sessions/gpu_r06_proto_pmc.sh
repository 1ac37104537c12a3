# PMC passes on the matrix-core MID prototype: where do its cycles go?
set -u
OUT=gpurun_out/r06proto; mkdir -p $OUT; export TMPDIR=/tmp; R=$(pwd)
BIN=$R/fastecc_amd/lib/proto_mid_mfma
( cd /tmp && rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA -d $R/$OUT/p1 -o pmc --output-format csv -- $BIN 19 1024 3 ) > $OUT/p1.log 2>&1
( cd /tmp && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $R/$OUT/p2 -o pmc --output-format csv -- $BIN 19 1024 3 ) > $OUT/p2.log 2>&1
( cd /tmp && rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM -d $R/$OUT/p3 -o pmc --output-format csv -- $BIN 19 1024 3 ) > $OUT/p3.log 2>&1
( cd /tmp && rocprofv3 --kernel-trace --stats -d $R/$OUT/st -o st --output-format csv -- $BIN 19 1024 3 ) > $OUT/st.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "mid9_mfma" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
c = {k: sum(v) / len(v) for k, v in agg.items()}
ms = None
for f in glob.glob(out + "/st/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mid9_mfma" in r["Name"]: ms = float(r["AverageNs"]) / 1e6
cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8
res = {"counters": c, "avg_ms": ms, "cycles": cyc, "clock_GHz": cyc / ms / 1e6 if ms else None}
if cyc:
    res["valu_per_simd_cycle"] = c["SQ_INSTS_VALU"] / 1024 / cyc
    res["mfma_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc
    wc = c["SQ_WAVE_CYCLES"]
    res["wave_cycle_split"] = {k: c[k] / wc for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")}
json.dump(res, open(out + "/pmc_proto_mid_mfma.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
