# The matrix-core direct pass under back-to-back launches: duration (kernel trace) and shader cycles (GRBM_GUI_ACTIVE / 8) of ten consecutive launches of the
# k + 64 encode at k = 2^19 x 4 KB — is the slow-down of later launches a clock drop (same cycles) or more cycles?
set -u
OUT=gpurun_out/r06dpow; mkdir -p $OUT; export TMPDIR=/tmp; R=$(pwd)
cat > /tmp/dpow.py <<'PY'
import os, sys, time
sys.path.insert(0, os.environ["REPO"])
import torch
import fastecc_amd as fe
k = 1 << 19
g = torch.Generator(device="cuda:0").manual_seed(1)
d = torch.randint(0, 0xFFF00001, (k * 1024,), generator=g, device="cuda:0", dtype=torch.int64).to(torch.int32)
q = torch.empty(64 * 1024, dtype=torch.int32, device="cuda:0")
with fe.Encoder(k + 64, k, 4096) as enc:
    enc.encode(d, q)
    torch.cuda.synchronize()
    time.sleep(0.5)
    for _ in range(10):
        enc.encode(d, q)
    torch.cuda.synchronize()
    time.sleep(0.5)
    for _ in range(3):
        enc.encode(d, q)
        torch.cuda.synchronize()
        time.sleep(0.3)
PY
( cd /tmp && REPO=$R rocprofv3 --kernel-trace -d "$R/$OUT/kt" -o kt --output-format csv -- python /tmp/dpow.py ) > "$OUT/kt.log" 2>&1
( cd /tmp && REPO=$R rocprofv3 --pmc GRBM_GUI_ACTIVE -d "$R/$OUT/pmc" -o pmc --output-format csv -- python /tmp/dpow.py ) > "$OUT/pmc.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, json
out = sys.argv[1]
dur, cyc = [], []
rows = []
for f in glob.glob(out + "/kt/**/*kernel_trace.csv", recursive=True): rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1) for r in rows if "direct_mfma_kernel" in r["Kernel_Name"]]
rows = []
for f in glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True): rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
cyc = [float(r["Counter_Value"]) / 8 for r in rows if "direct_mfma_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE"]
res = {"what": "direct_mfma_kernel<8> (64 outputs, k = 2^19 x 4 KB): launch 0 warms up, 1-10 back to back after 0.5 s idle, 11-13 one at a time with 0.3 s idle between; durations from a kernel-trace run, cycles from a separate GRBM_GUI_ACTIVE run of the same script",
       "duration_us": dur, "shader_cycles": [round(c) for c in cyc],
       "clock_GHz_if_the_two_runs_match": [round(c / (d * 1e3), 3) for c, d in zip(cyc, dur)] if len(cyc) == len(dur) else None}
json.dump(res, open(out + "/direct_power.json", "w"), indent=1)
print(json.dumps(res))
PY
