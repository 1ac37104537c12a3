# 64-bit field decode_prepare at k = 2^19, 2 % lost: phases of a cold process, then the kernels of the last (steady) call
set -u
OUT=gpurun_out/r06prep61; mkdir -p $OUT; export TMPDIR=/tmp; R=$(pwd)
{
echo "== FASTECC_TRACE_PREPARE=1 python tools/run_prepare.py 0.02 4 p61 (cold process) =="
FASTECC_TRACE_PREPARE=1 python tools/run_prepare.py 0.02 4 p61 2>&1 | grep -v amdgpu.ids
echo
echo "== kernels of the last (steady) call: rocprofv3 --kernel-trace =="
} > $OUT/prepare_trace_p61.txt
( cd /tmp && rocprofv3 --kernel-trace -d $R/$OUT/t -o t --output-format csv -- python $R/tools/run_prepare.py 0.02 3 p61 ) > $OUT/t.log 2>&1
python3 - $OUT >> $OUT/prepare_trace_p61.txt <<'PY'
import csv, glob, sys, re
out = sys.argv[1]
rows = []
for f in glob.glob(out + "/t/**/*kernel_trace.csv", recursive=True): rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
start = [i for i, r in enumerate(rows) if "k_roots" in r["Kernel_Name"]][-1]
while start > 0 and any(t in rows[start - 1]["Kernel_Name"] for t in ("k_erased_list", "fillBuffer", "k_mark_unused", "copyBuffer")): start -= 1
last = rows[start:]
t0 = int(last[0]["Start_Timestamp"])
print("kernels in the last prepare:", len(last), "span ms", (int(last[-1]["End_Timestamp"]) - t0) / 1e6, "busy ms", sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last) / 1e6)
for r in last:
    n = re.sub(r"fastecc::p61::|\(anonymous namespace\)::|void ", "", r["Kernel_Name"])[:100]
    print("%8.1f us + %6.1f us  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, n))
PY
tail -5 $OUT/prepare_trace_p61.txt
