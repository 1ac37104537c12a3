#!/bin/bash
# Timeline of one FASTECC_MEM_HOST_PINNED encode: rocprofv3 kernel + memory-copy trace -> copies and kernels in time order
set -u
OUT=${1:-gpurun_out/trace_host}; mkdir -p "$OUT"; export TMPDIR=/tmp
R=$(pwd)
cat > /tmp/host_once.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["R"])
import torch, fastecc_amd as fe
N, S = 1 << 19, 1024
hx = torch.zeros(N * S, dtype=torch.int32).pin_memory(); hp = torch.empty(N * S, dtype=torch.int32).pin_memory()
with fe.Encoder(2 * N, N, 4 * S) as enc:
    enc.set_option("host_slabs", int(os.environ.get("SLABS", "8")))
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        enc.encode(hx.data_ptr(), hp.data_ptr(), stream=st, mem=fe.MEM_HOST_PINNED); torch.cuda.synchronize()
PY
( cd /tmp && R=$R rocprofv3 --kernel-trace --memory-copy-trace -d "$R/$OUT" -o t --output-format csv -- python /tmp/host_once.py ) > "$OUT/run.log" 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
ev = []
for f in glob.glob(out + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", r.get("Kind", "?")) + " " + r.get("Bytes", r.get("Size", "?"))))
for f in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "ntt_tile" in r["Kernel_Name"]:
            n = r["Kernel_Name"]; ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "kernel " + n[n.index("<"):n.index(">") + 1]))
ev.sort()
# the last encode: events after the last big gap
big = [e for e in ev if e[2].startswith("copy")]
t_last = ev[-1][1]
sel = [e for e in ev if e[0] > t_last - 80_000_000]
t0 = sel[0][0]
for s, e, n in sel:
    print("%9.2f ms .. %9.2f ms  (%7.2f)  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, n))
PY
