#!/usr/bin/env python3
"""Column-slab scheduling x cache policy sweep for the headline encode (MI355X memory-side cache experiment).
One JSON line per configuration: ms per (2^20, 2^19) x 4 KB encode, out of place and in place."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import fastecc_amd  # noqa: E402

log2k = int(sys.argv[1]) if len(sys.argv) > 1 else 19
k, bb = 1 << log2k, 4096
S = bb // 4
data = torch.randint(0, 0xFFF00001, (k * S,), dtype=torch.int64, device="cuda:0").to(torch.int32)
parity = torch.empty_like(data)
enc = fastecc_amd.Encoder(2 * k, k, bb)
stream = torch.cuda.current_stream().cuda_stream


def timed(fn, steps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


enc.encode(data, parity, stream=stream)
torch.cuda.synchronize()
want = parity.clone()
for slabs, mode in ((1, 0), (4, 1), (8, 1), (16, 1), (32, 1), (8, 0), (16, 0), (32, 0)):
    for policy in (15, 0, 3, 12, 1, 2, 13, 14):
        enc.set_option("slabs", slabs)
        enc.set_option("slab_mode", mode)
        enc.set_option("cache_policy", policy)
        parity.zero_()
        ms = timed(lambda: enc.encode(data, parity, stream=stream))
        ok = bool(torch.equal(parity, want))
        work = data.clone()
        ms_in = timed(lambda: enc.encode(work, work, stream=stream), steps=4)
        print(json.dumps({"slabs": slabs, "mode": "sequential" if mode else "staggered", "cache_policy": policy, "ms": round(ms, 4),
                          "GBps": round(2.0 * k * bb / ms / 1e6, 1), "ms_in_place": round(ms_in, 4), "parity_ok": ok}), flush=True)
