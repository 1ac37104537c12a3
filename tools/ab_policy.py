#!/usr/bin/env python3
"""A/B of cache_policy values for the headline encode: interleaved rounds of 40 encodes each, median per policy."""
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import fastecc_amd  # noqa: E402

k, bb = 1 << 19, 4096
data = torch.randint(0, 0xFFF00001, (k * bb // 4,), dtype=torch.int64, device="cuda:0").to(torch.int32)
parity = torch.empty_like(data)
enc = fastecc_amd.Encoder(2 * k, k, bb)
stream = torch.cuda.current_stream().cuda_stream
policies = [int(p) for p in sys.argv[1:]] or [15, 3, 13, 7, 11]
res = {p: [] for p in policies}
for rnd in range(6):
    for p in policies:
        enc.set_option("cache_policy", p)
        for _ in range(3):
            enc.encode(data, parity, stream=stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            enc.encode(data, parity, stream=stream)
        torch.cuda.synchronize()
        res[p].append((time.perf_counter() - t0) / 40 * 1e3)
print(json.dumps({str(p): {"median_ms": round(statistics.median(v), 4), "min_ms": round(min(v), 4), "max_ms": round(max(v), 4)} for p, v in res.items()}))
