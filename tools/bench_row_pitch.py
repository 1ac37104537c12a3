#!/usr/bin/env python3
"""Does the power-of-two block stride cost the outer passes anything?  The headline encode (k = 2^19, 4096-byte blocks) on stripes whose rows are
4096 B of data at a pitch of 4096, 4224, 4352, 4608 or 5120 bytes (fastecc_set_option "row_pitch_words"): per-kernel averages (HIP events around
every launch) and the encode time, interleaved rounds, best per pitch.  Tiles of the outer passes touch 1024 rows 512 blocks apart: with the plain
layout that is a stride of exactly 2 MiB."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import fastecc_amd as fe  # noqa: E402

k, S = 1 << 19, 1024
st = torch.cuda.current_stream().cuda_stream
PITCHES = tuple(int(v) for v in sys.argv[1].split(",")) if len(sys.argv) > 1 else (1024, 1056, 1088, 1152, 1280)


def ev(fn, reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


plain = torch.randint(0, 0xFFF00001, (k, S), dtype=torch.int64, device="cuda:0").to(torch.int32)
want = torch.empty_like(plain)
with fe.Encoder(2 * k, k, 4 * S) as e0:
    e0.encode(plain.view(-1), want.view(-1), stream=st)
torch.cuda.synchronize()
ctx = {}
for L in PITCHES:
    enc = fe.Encoder(2 * k, k, 4 * S)
    if L != S:
        enc.set_option("row_pitch_words", L)
    d = torch.zeros((k, L), dtype=torch.int32, device="cuda:0")
    d[:, :S] = plain
    p = torch.zeros((k, L), dtype=torch.int32, device="cuda:0")
    for _ in range(3):
        enc.encode(d.view(-1), p.view(-1), stream=st)
    torch.cuda.synchronize()
    ctx[L] = (enc, d, p, bool(torch.equal(p[:, :S], want)))
best = {L: 1e9 for L in PITCHES}
for _ in range(3):
    for L in PITCHES:
        enc, d, p, _ = ctx[L]
        best[L] = min(best[L], ev(lambda: enc.encode(d.view(-1), p.view(-1), stream=st), 20))
for L in PITCHES:
    enc, d, p, same = ctx[L]
    enc.profile(True)
    enc.profile_reset()
    for _ in range(10):
        enc.encode(d.view(-1), p.view(-1), stream=st)
    kern = {kn: round(v[0] / v[1], 4) for kn, v in enc.profile_read().items()}
    enc.profile(False)
    print(json.dumps({"row_pitch_bytes": 4 * L, "ms": round(best[L], 4), "same_parity": same, "plan": enc.plan(), "kernels_avg_ms": kern}), flush=True)
    enc.close()
