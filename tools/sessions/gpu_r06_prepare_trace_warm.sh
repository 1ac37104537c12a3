set -u
OUT=gpurun_out/r06prepw; mkdir -p $OUT; export TMPDIR=/tmp; R=$(pwd)
FASTECC_TRACE_PREPARE=1 python tools/run_prepare.py 0.02 3 warm > $OUT/trace.txt 2>&1
( cd /tmp && rocprofv3 --hip-trace --kernel-trace -d $R/$OUT/t -o t --output-format csv -- python $R/tools/run_prepare.py 0.02 3 warm ) > $OUT/t.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys
out = sys.argv[1]
rows = []
for f in glob.glob(out + "/t/**/*kernel_trace.csv", recursive=True): rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "presence_counts" in r["Kernel_Name"]]
first = rows[idx[0]:idx[1]]
t0 = int(first[0]["Start_Timestamp"]); prev = t0
print(len(first), "kernels in the first prepare; busy ms", sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in first) / 1e6)
for r in first[:-1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if (s - prev) / 1e3 > 40: print("%9.1f us  gap %8.1f us before %s" % ((s - t0) / 1e3, (s - prev) / 1e3, r["Kernel_Name"][:90]))
    prev = e
api = []
for f in glob.glob(out + "/t/**/*hip_api_trace.csv", recursive=True): api += list(csv.DictReader(open(f)))
api.sort(key=lambda r: int(r["Start_Timestamp"]))
t1 = int(rows[idx[1]]["Start_Timestamp"])
w = [r for r in api if t0 - 300000 <= int(r["Start_Timestamp"]) <= t1]
import collections
agg = collections.defaultdict(lambda: [0, 0])
for r in w:
    agg[r["Function"]][0] += 1; agg[r["Function"]][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:10]: print("%-26s n=%4d total %.3f ms" % (k, n, d / 1e6))
slow = sorted(w, key=lambda r: int(r["Start_Timestamp"]) - int(r["End_Timestamp"]))[:8]
for r in slow: print("  %.1f us %s at %.1f us" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Function"], (int(r["Start_Timestamp"]) - t0) / 1e3))
PY
head -12 $OUT/trace.txt
