set -u
OUT=gpurun_out/r06prep; mkdir -p $OUT; export TMPDIR=/tmp; R=$(pwd)
FASTECC_TRACE_PREPARE=1 python tools/run_prepare.py 0.02 6 > $OUT/trace_002.txt 2>&1
( cd /tmp && rocprofv3 --kernel-trace --stats -d $R/$OUT/st -o st --output-format csv -- python $R/tools/run_prepare.py 0.02 6 ) > $OUT/st.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys
out = sys.argv[1]
rows = []
for f in glob.glob(out + "/st/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last prepare call: kernels after the last erased_list_kernel
idx = max(i for i, r in enumerate(rows) if "erased_list_kernel" in r["Kernel_Name"])
last = rows[idx - 2:]
t0 = int(last[0]["Start_Timestamp"])
print("kernels in the last prepare:", len(last), "span ms", (int(last[-1]["End_Timestamp"]) - t0) / 1e6, "busy ms", sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last) / 1e6)
for r in last:
    print("%8.1f us +%7.1f us  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:110]))
PY
