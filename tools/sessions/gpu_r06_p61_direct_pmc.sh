# Counters of the 64-bit field's direct pass (k_direct_accumulate) at k = 2^18 x 64 KB, 8 and 16 data blocks lost: VALU instructions per cycle and SIMD
# against the headline yardstick, one counter group per run (no tracing with --pmc)
set -u
OUT=gpurun_out/r06p61d; mkdir -p $OUT; export TMPDIR=/tmp; R=$(pwd)
cat > /tmp/p61d.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["REPO"])
import numpy as np, torch
import fastecc_amd as fe
from bench_common import random_stripe_p61
dev = torch.device("cuda", 0)
k, bb = 1 << 18, 65536
data = random_stripe_p61(k * (bb // 8), dev, seed=0x61A)
parity = torch.empty_like(data)
with fe.Encoder(2 * k, k, bb, field=fe.FIELD_GF_P61_SQUARED) as enc:
    enc.encode(data, parity)
    for e in (8, 16):
        dp, pp = np.ones(k, np.uint8), np.ones(k, np.uint8)
        dp[np.random.default_rng(e).permutation(k)[:e]] = 0
        enc.decode_prepare(dp, pp)
        enc.decode(data, parity)
        torch.cuda.synchronize()
PY
run() { local name=$1; shift
  ( cd /tmp && REPO=$R rocprofv3 --pmc "$@" -d "$R/$OUT/$name" -o pmc --output-format csv -- python /tmp/p61d.py ) > "$OUT/$name.log" 2>&1
}
run sq1 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
run sq2 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA
run grbm GRBM_GUI_ACTIVE
( cd /tmp && REPO=$R rocprofv3 --kernel-trace -d "$R/$OUT/kt" -o kt --output-format csv -- python /tmp/p61d.py ) > "$OUT/kt.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "k_direct_accumulate" not in k: continue
        agg["k_direct_accumulate"][row["Counter_Name"]].append(float(row["Counter_Value"]))
dur = []
for f in glob.glob(out + "/kt/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_direct_accumulate" in row["Kernel_Name"]: dur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
res = {"launches_in_order": "8 lost (1 sweep of 8 outputs), 16 lost (2 sweeps)", "duration_ms": dur,
       "counters_per_launch": {c: v for c, v in agg["k_direct_accumulate"].items()}}
c = res["counters_per_launch"]
if "GRBM_GUI_ACTIVE" in c and "SQ_INSTS_VALU" in c:
    res["valu_per_cycle_and_simd"] = [round(i / (1024 * g / 8), 4) for i, g in zip(c["SQ_INSTS_VALU"], c["GRBM_GUI_ACTIVE"])]
    res["clock_GHz"] = [round(g / 8 / (d * 1e6), 3) for g, d in zip(c["GRBM_GUI_ACTIVE"], dur)] if len(dur) == len(c["GRBM_GUI_ACTIVE"]) else None
    res["yardstick"] = "the isolated radix-2 butterfly loop issues 0.307 VALU instructions per cycle and SIMD (profiles/r06/pmc_valu_default_plan.json)"
json.dump(res, open(out + "/pmc_p61_direct.json", "w"), indent=1)
print(json.dumps(res)[:1500])
PY
