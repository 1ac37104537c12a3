#!/bin/bash
# Round 4, session E: the small form of the split decoder — tests, then decode / repair timings against the block-group form
set -u
TAG=${1:-r04e}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_decode.py tests/test_gpu_general.py tests/test_gpu_fuzz.py tests/test_gpu_sharded.py -m gpu -x -q > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest.log"
python - > "$OUT/decode_forms.jsonl" <<'PY'
import json, time, numpy as np, torch, sys, os
sys.path.insert(0, os.getcwd())
import fastecc_amd as fe
k, S = 1 << 19, 1024
data = torch.randint(0, 0xFFF00001, (k * S,), dtype=torch.int64, device="cuda:0").to(torch.int32)
parity = torch.empty_like(data)
st = torch.cuda.current_stream().cuda_stream
def ev(fn, reps=5):
    fn(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
with fe.Encoder(2 * k, k, 4 * S) as enc:
    enc.encode(data, parity, stream=st); torch.cuda.synchronize()
    rng = np.random.default_rng(7)
    for name, frac in (("0.1 %", 0.001), ("2 %", 0.02), ("10 %", 0.10), ("25 %", 0.25), ("40 %", 0.40), ("50 %", 0.50)):
        lost = rng.permutation(2 * k)[: int(2 * k * frac)]
        dp, pp = np.ones(k, np.uint8), np.ones(k, np.uint8)
        dp[lost[lost < k]] = 0; pp[lost[lost >= k] - k] = 0
        di = torch.from_numpy(np.flatnonzero(dp == 0)).to("cuda:0"); pi = torch.from_numpy(np.flatnonzero(pp == 0)).to("cuda:0")
        sd, sp = data.view(k, S)[di].clone(), parity.view(k, S)[pi].clone()
        row = {"lost": name}
        for form, opt in (("small", 1), ("groups", 2)):
            enc.set_option("decode_split", opt)
            t0 = time.perf_counter(); enc.decode_prepare(dp, pp); first = (time.perf_counter() - t0) * 1e3
            t0 = time.perf_counter(); enc.decode_prepare(dp, pp); steady = (time.perf_counter() - t0) * 1e3
            data.view(k, S)[di] = -1; parity.view(k, S)[pi] = -2
            enc.repair(data, parity, stream=st); torch.cuda.synchronize()
            ok = bool(torch.equal(data.view(k, S)[di], sd)) and bool(torch.equal(parity.view(k, S)[pi], sp))
            enc.profile(True); enc.profile_reset(); enc.decode(data, parity, stream=st); kern = {n: round(v[0] / v[1], 3) for n, v in enc.profile_read().items()}; enc.profile(False)
            row[form] = {"prepare_first_ms": round(first, 2), "prepare_ms": round(steady, 2), "decode_ms": round(ev(lambda: enc.decode(data, parity, stream=st)), 3),
                         "repair_ms": round(ev(lambda: enc.repair(data, parity, stream=st)), 3), "restored": ok, "kernels": kern}
        print(json.dumps(row), flush=True)
PY
python - "$OUT/decode_forms.jsonl" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    r = json.loads(l)
    print(r["lost"], {f: (r[f]["decode_ms"], r[f]["repair_ms"], r[f]["prepare_ms"], r[f]["restored"]) for f in ("small", "groups")})
    print("     ", r["small"]["kernels"])
PY
