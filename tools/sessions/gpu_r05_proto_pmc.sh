set -u
OUT=gpurun_out/r05f64; mkdir -p $OUT; export TMPDIR=/tmp; R=$(pwd)
( cd /tmp && rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $R/$OUT/pmc_proto -o pmc --output-format csv -- $R/fastecc_amd/lib/proto_mid_f64 19 1024 ) > $OUT/pmc_proto.log 2>&1
( cd /tmp && rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY -d $R/$OUT/pmc_proto2 -o pmc --output-format csv -- $R/fastecc_amd/lib/proto_mid_f64 19 1024 ) > $OUT/pmc_proto2.log 2>&1
( cd /tmp && rocprofv3 --kernel-trace --stats -d $R/$OUT/stats_proto -o st --output-format csv -- $R/fastecc_amd/lib/proto_mid_f64 19 1024 ) > $OUT/stats_proto.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(out + "/pmc_proto*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mid9_f64" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
c = {k: sum(v) / len(v) for k, v in agg.items()}
print(c)
cyc = c["GRBM_GUI_ACTIVE"] / 8
print("cycles", cyc, "valu/simd/cycle", c["SQ_INSTS_VALU"] / 1024 / cyc)
for f in glob.glob(out + "/stats_proto/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mid9_f64" in r["Name"]:
            ms = float(r["AverageNs"]) / 1e6; print("avg ms", ms, "clock GHz", cyc / ms / 1e6)
PY
