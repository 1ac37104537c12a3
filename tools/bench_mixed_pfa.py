#!/usr/bin/env python3
"""Composite odd factors (FASTECC_CODE_MIXED_RADIX_PFA) against the next order of the seven plain factors (FASTECC_CODE_MIXED_RADIX) and
against zero extension to a power of two: k = q * 2^m data blocks of 4 KB, n = 2k.  Three different codes over the same data; the question
is what a host pays per data + parity byte of ITS stripe (the padding of the larger orders is the host's loss)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import fastecc_amd as fe  # noqa: E402

P = 0xFFF00001
S = 1024
cases = [(21, 14), (21, 16), (35, 13), (35, 15), (39, 13), (45, 13), (45, 15), (63, 12), (63, 14), (63, 15), (65, 10), (65, 13), (91, 10), (105, 10), (117, 10), (117, 12)]
if len(sys.argv) > 1:
    cases = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for q, m in cases:
    k = q << m
    data = torch.randint(0, P, (k * S,), dtype=torch.int64, device="cuda:0").to(torch.int32)
    parity = torch.empty_like(data)
    row = {"k": k, "q": q, "m": m, "block_bytes": 4 * S}
    for name, flags in (("pfa", fe.CODE_MIXED_RADIX_PFA), ("pfa_not_fused", fe.CODE_MIXED_RADIX_PFA), ("plain_factors", fe.CODE_MIXED_RADIX), ("zero_extended_pow2", 0)):
        try:
            enc = fe.Encoder(2 * k, k, 4 * S, flags=flags)
        except fe.FastEccError as e:
            row[name] = {"unsupported": str(e)}
            continue
        with enc:
            if name == "pfa_not_fused":
                if "+" not in enc.plan():
                    continue
                enc.set_option("fuse_radix", 0)
            stream = torch.cuda.current_stream().cuda_stream
            for _ in range(3):
                enc.encode(data, parity, stream=stream)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                enc.encode(data, parity, stream=stream)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 10 * 1e3
            enc.profile(True)
            enc.profile_reset()
            for _ in range(3):
                enc.encode(data, parity, stream=stream)
            kern = {kn: round(v[0] / v[1], 4) for kn, v in enc.profile_read().items()}
            enc.profile(False)
            order = fe.mixed_radix_order(k, pfa=flags == fe.CODE_MIXED_RADIX_PFA) if flags else 1 << (k - 1).bit_length()
            row[name] = {"order": order, "ms": round(ms, 4), "GBps": round(2.0 * k * 4 * S / (ms * 1e-3) / 1e9, 1), "plan": enc.plan(), "kernel_ms": kern}
    print(json.dumps(row), flush=True)
    del data, parity
    torch.cuda.empty_cache()
