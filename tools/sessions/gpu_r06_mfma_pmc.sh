# PMC passes on the matrix-core stage microbenchmark (2 waves per SIMD): cycles, VALU and MFMA activity per probe kernel.
set -u
OUT=gpurun_out/r06pmc; mkdir -p $OUT; export TMPDIR=/tmp; R=$(pwd)
BIN=$R/fastecc_amd/lib/microbench_mfma_dft
( cd /tmp && rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA -d $R/$OUT/p1 -o pmc --output-format csv -- $BIN 2 ) > $OUT/p1.log 2>&1
( cd /tmp && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC -d $R/$OUT/p2 -o pmc --output-format csv -- $BIN 2 ) > $OUT/p2.log 2>&1
( cd /tmp && rocprofv3 --kernel-trace --stats -d $R/$OUT/st -o st --output-format csv -- $BIN 2 ) > $OUT/st.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
# dispatches in order: group by kernel name, keep the LAST 3 dispatches of each (the timed ones: iters = 2048 / 1024)
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        rows[r["Kernel_Name"]][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
res = {}
for k, cs in rows.items():
    d = {}
    for c, v in cs.items():
        v.sort()
        # per kernel name there are several probes (occupancy 8 / 2 for valu5; one per template instance otherwise); keep dispatch groups
        d[c] = [x[1] for x in v]
    res[k] = d
json.dump(res, open(out + "/pmc_microbench_mfma_dft_raw.json", "w"), indent=0)
for k, d in res.items():
    print(k[:90])
    for c, v in sorted(d.items()): print("   %-28s %s" % (c, " ".join("%.4g" % x for x in v[-4:])))
PY
