set -u
OUT=gpurun_out/r06prep61; mkdir -p $OUT; export TMPDIR=/tmp; R=$(pwd)
( cd /tmp && rocprofv3 --kernel-trace -d $R/$OUT/t -o t --output-format csv -- python $R/tools/run_prepare.py 0.02 3 p61 ) > $OUT/t.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
rows = []
for f in glob.glob(out + "/t/**/*kernel_trace.csv", recursive=True): rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_roots" in r["Kernel_Name"]]
last = rows[idx[-1]:]
t0 = int(last[0]["Start_Timestamp"])
print(len(last), "kernels; span ms", (int(last[-1]["End_Timestamp"]) - t0) / 1e6, "busy", sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last) / 1e6)
agg = collections.defaultdict(lambda: [0, 0])
for r in last:
    n = r["Kernel_Name"].split("(")[0][-60:]
    agg[n][0] += 1; agg[n][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]): print("%4d x %8.1f us  %s" % (n, d / 1e3, k))
PY
