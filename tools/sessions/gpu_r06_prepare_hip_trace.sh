set -u
OUT=gpurun_out/r06prep2; mkdir -p $OUT; export TMPDIR=/tmp; R=$(pwd)
( cd /tmp && rocprofv3 --hip-trace --kernel-trace -d $R/$OUT/t -o t --output-format csv -- python $R/tools/run_prepare.py 0.02 2 ) > $OUT/t.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
api = []
for f in glob.glob(out + "/t/**/*hip_api_trace.csv", recursive=True):
    api += list(csv.DictReader(open(f)))
api.sort(key=lambda r: int(r["Start_Timestamp"]))
# find the window of the first prepare: from the first hipMemcpyAsync H2D of size N after encoder creation... simply: take calls between the first and second presence_counts launch
names = [r["Function"] for r in api]
marks = [i for i, r in enumerate(api) if r["Function"] == "hipMemsetAsync"]
# crude: first prepare = calls from first 'hipMemcpyAsync' to the second occurrence of a D2H hipMemcpy of the counts; print aggregate by function over the whole run instead
agg = collections.defaultdict(lambda: [0, 0])
for r in api:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg[r["Function"]][0] += 1
    agg[r["Function"]][1] += d
for k, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print("%-28s n=%5d total %.3f ms avg %.1f us" % (k, n, d / 1e6, d / 1e3 / n))
PY
