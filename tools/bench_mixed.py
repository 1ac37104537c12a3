#!/usr/bin/env python3
"""Mixed-radix orders against zero extension (profiles/r02/mixed_radix_bench.json): k = q * 2^m data blocks of 4 KB,
n = 2k, encoded (a) on transform order q * 2^m (FASTECC_CODE_MIXED_RADIX: two odd-radix passes + the power-of-two pipeline
on q stripes) and (b) zero-extended to the next power of two (fastecc_create).  The two are different codes over the same
data; the question is what a host pays per data+parity byte."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import fastecc_amd as fe  # noqa: E402

P = 0xFFF00001
S = 1024
rows = []
for q, m in ((3, 17), (5, 16), (7, 16), (9, 15), (13, 15), (15, 15), (3, 10), (9, 16), (15, 19)):
    k = q << m
    data = torch.randint(0, P, (k * S,), dtype=torch.int64, device="cuda:0").to(torch.int32)
    parity = torch.empty_like(data)
    row = {"k": k, "q": q, "m": m, "block_bytes": 4 * S}
    for name, flags in (("mixed_radix", fe.CODE_MIXED_RADIX), ("zero_extended_pow2", 0)):
        try:
            enc = fe.Encoder(2 * k, k, 4 * S, flags=flags)
        except fe.FastEccError as e:
            row[name] = {"unsupported": str(e)}  # k > 2^19 has no power-of-two order with a root of order 2N
            continue
        with enc:
            stream = torch.cuda.current_stream().cuda_stream
            for _ in range(3):
                enc.encode(data, parity, stream=stream)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                enc.encode(data, parity, stream=stream)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 20 * 1e3
            enc.profile(True)
            enc.profile_reset()
            for _ in range(5):
                enc.encode(data, parity, stream=stream)
            kern = {kn: round(v[0] / v[1], 4) for kn, v in enc.profile_read().items()}
            enc.profile(False)
            row[name] = {"ms": round(ms, 4), "GBps": round(2.0 * k * 4 * S / (ms * 1e-3) / 1e9, 1), "plan": enc.plan(), "kernel_ms": kern}
    rows.append(row)
    print(json.dumps(row), flush=True)
