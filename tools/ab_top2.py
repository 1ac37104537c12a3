#!/usr/bin/env python3
"""A/B: the headline encode through tile_kernels.hip's outer tiles (fastecc_create) against the same code with its top level
handled as a radix-2 "odd" level fused with the next 8 levels in mixed_kernels.hip (FASTECC_CODE_TOP_RADIX2).  Interleaved
rounds; prints medians, per-kernel times, and whether the two parities are identical."""
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import fastecc_amd as fe  # noqa: E402

log2k = int(sys.argv[1]) if len(sys.argv) > 1 else 19
k, bb = 1 << log2k, 4096
data = torch.randint(0, 0xFFF00001, (k * bb // 4,), dtype=torch.int64, device="cuda:0").to(torch.int32)
pa, pb = torch.empty_like(data), torch.empty_like(data)
a = fe.Encoder(2 * k, k, bb)
b = fe.Encoder(2 * k, k, bb, flags=fe.CODE_TOP_RADIX2)
stream = torch.cuda.current_stream().cuda_stream
a.encode(data, pa, stream=stream)
b.encode(data, pb, stream=stream)
torch.cuda.synchronize()
out = {"identical": bool(torch.equal(pa, pb)), "plan_a": a.plan(), "plan_b": b.plan()}
res = {"a": [], "b": []}
for rnd in range(6):
    for name, enc, par in (("a", a, pa), ("b", b, pb)):
        for _ in range(3):
            enc.encode(data, par, stream=stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            enc.encode(data, par, stream=stream)
        torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) / 40 * 1e3)
for name, enc, par in (("a", a, pa), ("b", b, pb)):
    enc.profile(True)
    enc.profile_reset()
    for _ in range(10):
        enc.encode(data, par, stream=stream)
    out["kernels_" + name] = {kn: round(v[0] / v[1], 4) for kn, v in enc.profile_read().items()}
    out["median_ms_" + name] = round(statistics.median(res[name]), 4)
print(json.dumps(out))
