#!/bin/bash
# Samples rocm-smi power / clocks while the headline encode (or one kernel mix) runs in a loop.  usage: tools/power_probe.sh <outdir>
OUT=${1:-gpurun_out/power}; mkdir -p "$OUT"
python - <<'PY' > "$OUT/loop.log" 2>&1 &
import sys, time, os
sys.path.insert(0, os.getcwd())
import torch, fastecc_amd
k, bb = 1 << 19, 4096
data = torch.randint(0, 0xFFF00001, (k * bb // 4,), dtype=torch.int64, device="cuda:0").to(torch.int32)
par = torch.empty_like(data)
enc = fastecc_amd.Encoder(2 * k, k, bb)
st = torch.cuda.current_stream().cuda_stream
t_end = time.time() + 14
n = 0
torch.cuda.synchronize(); t0 = time.time()
while time.time() < t_end:
    for _ in range(50):
        enc.encode(data, par, stream=st)
    torch.cuda.synchronize(); n += 50
print("encodes", n, "ms_per_encode", (time.time() - t0) / n * 1e3)
PY
LOOP=$!
sleep 6   # import + allocation
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (junction|edge)" | tr -s ' ' | head -8
  echo "--"
  sleep 1
done > "$OUT/smi_under_load.txt"
wait $LOOP
cat "$OUT/loop.log" | tail -1
rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | tr -s ' ' > "$OUT/smi_idle.txt"
rocm-smi --showmaxpower 2>/dev/null | grep -i "power" | tr -s ' ' >> "$OUT/smi_idle.txt"
echo "== under load"; cat "$OUT/smi_under_load.txt" | head -24; echo "== idle"; cat "$OUT/smi_idle.txt"
