#!/usr/bin/env python3
"""Encode time for (n,k) that are not powers of two (zero extension, include/fastecc.h), 4 KB blocks, HBM-resident."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import fastecc_amd as fe  # noqa: E402

P = 0xFFF00001
S = 1024
out = []
for k, m in ((400000, 100000), (300000, 300000), (524288, 100000), (100000, 20000), (10000, 2000)):
    g = torch.Generator(device="cuda:0").manual_seed(k)
    data = torch.randint(0, P, (k * S,), dtype=torch.int64, device="cuda:0", generator=g).to(torch.int32)
    parity = torch.empty(m * S, dtype=torch.int32, device="cuda:0")
    with fe.Encoder(k + m, k, 4 * S) as enc:
        for _ in range(3):
            enc.encode(data, parity)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            enc.encode(data, parity)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        out.append({"k": k, "parity": m, "ms": round(ms, 3), "data_plus_parity_GBps": round((k + m) * S * 4 / (ms * 1e-3) / 1e9, 1), "plan": enc.plan()})
print(json.dumps(out))
