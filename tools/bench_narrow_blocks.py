#!/usr/bin/env python3
"""Encode time by block size at k = 2^19 (the sub-slab sizes of ONE stripe spread over 2 .. 8 GPUs: 4096 / G / sub_slabs bytes), default plan and the
explicit plan ids, as 2 GiB-stripe equivalents (ms * 4096 / block_bytes).  One JSON line per block size; every result is checked against plan 3100's."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import fastecc_amd as fe  # noqa: E402

k = 1 << 19
st = torch.cuda.current_stream().cuda_stream


def ev(fn, reps=20):
    fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


# optional arguments: block sizes and plan ids, comma separated (3100 is always measured: the others' results are compared with its parity)
SIZES = tuple(int(v) for v in sys.argv[1].split(",")) if len(sys.argv) > 1 else (128, 256, 512, 1024, 2048, 4096)
PLANS = tuple(dict.fromkeys([int(v) for v in sys.argv[2].split(",")] + [3100])) if len(sys.argv) > 2 else (0, 3100, 3090, 4090, 4100)
PROFILE = len(sys.argv) > 3  # any third argument: per-kernel averages (HIP events around every launch) of 10 more encodes per plan
for bb in SIZES:
    S = bb // 4
    d = torch.randint(0, 0xFFF00001, (k * S,), dtype=torch.int64, device="cuda:0").to(torch.int32)
    ref, p = torch.empty_like(d), torch.empty_like(d)
    row = {"block_bytes": bb, "k": "2^19", "how": "every plan warmed 5 times, then 3 rounds over the plans of 30 encodes each (HIP events), the best round per plan"}
    encs = {}
    for plan in PLANS:
        encs[plan] = fe.Encoder(2 * k, k, bb)
        if plan:
            encs[plan].set_plan(plan)
    encs[3100].encode(d, ref, stream=st)
    same = {}
    for plan in PLANS:
        for _ in range(5):
            encs[plan].encode(d, p, stream=st)
        torch.cuda.synchronize()
        same[plan] = bool(torch.equal(p, ref))
    best = {plan: 1e9 for plan in PLANS}
    for _ in range(3):
        for plan in PLANS:
            best[plan] = min(best[plan], ev(lambda: encs[plan].encode(d, p, stream=st), 30))
    for plan in PLANS:
        row["plan %d" % plan] = {"ms": round(best[plan], 4), "ms_per_2GiB_equivalent": round(best[plan] * 4096 / bb, 3), "plan": encs[plan].plan(), "same_parity": same[plan]}
        if PROFILE:
            encs[plan].profile(True)
            encs[plan].profile_reset()
            for _ in range(10):
                encs[plan].encode(d, p, stream=st)
            row["plan %d" % plan]["kernels_avg_ms"] = {kn: round(v[0] / v[1], 4) for kn, v in encs[plan].profile_read().items()}
            encs[plan].profile(False)
        encs[plan].close()
    print(json.dumps(row), flush=True)
