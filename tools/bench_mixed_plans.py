#!/usr/bin/env python3
"""Mixed-radix orders q * 2^m x 4 KB under the default plan (MID10 between the fused outer passes) and with a shorter MID (plans 3090, 3080):
ms per encode, plan text, same parity.  The shorter MID loses here — the fused outer passes carry the odd-radix transform and are VALU-bound."""
import json, os, sys, time
sys.path.insert(0, "/root/repo")
import torch
import fastecc_amd as fe
P = 0xFFF00001
S = 1024
for q, m in ((3, 17), (5, 16), (7, 16), (9, 15), (13, 15), (15, 15), (21, 14), (9, 16), (3, 18)):
    k = q << m
    data = torch.randint(0, P, (k * S,), dtype=torch.int64, device="cuda:0").to(torch.int32)
    parity = torch.empty_like(data)
    row = {"q": q, "m": m}
    flags = fe.CODE_MIXED_RADIX_PFA if q > 15 else fe.CODE_MIXED_RADIX
    with fe.Encoder(2 * k, k, 4 * S, flags=flags) as enc:
        stream = torch.cuda.current_stream().cuda_stream
        ref = None
        for plan in (0, 3090, 3080, 0, 3090):
            try:
                enc.set_plan(plan)
            except fe.FastEccError as e:
                row[str(plan)] = "unsupported"
                continue
            for _ in range(3):
                enc.encode(data, parity, stream=stream)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                enc.encode(data, parity, stream=stream)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 10 * 1e3
            if ref is None:
                ref = parity.clone()
            row.setdefault(str(plan), []).append((round(ms, 4), enc.plan(), bool(torch.equal(parity, ref))))
    print(json.dumps(row), flush=True)
