#!/usr/bin/env python3
"""Codes with few parity blocks at k = 2^19 x 4 KB: the direct encoder (one read of the data, n - k accumulators per word) against the
transform pipeline (option encode_direct_max = 0).  One JSON line per m."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import fastecc_amd as fe  # noqa: E402

k, bb = 1 << 19, 4096
data = torch.randint(0, 0xFFF00001, (k * bb // 4,), dtype=torch.int64, device="cuda:0").to(torch.int32)
stream = torch.cuda.current_stream().cuda_stream
for m in (1, 2, 4, 8):
    par = {d: torch.empty(m * bb // 4, dtype=torch.int32, device="cuda:0") for d in (8, 0)}
    row = {"k": k, "m": m, "block_bytes": bb}
    with fe.Encoder(k + m, k, bb) as enc:
        for direct_max in (8, 0):
            enc.set_option("encode_direct_max", direct_max)
            for _ in range(3):
                enc.encode(data, par[direct_max], stream=stream)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                enc.encode(data, par[direct_max], stream=stream)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 20 * 1e3
            row["direct" if direct_max else "transform"] = {"ms": round(ms, 4), "data_GBps": round(k * bb / ms / 1e6, 1)}
        row["identical"] = bool(torch.equal(par[8], par[0]))
        row["transform_plan"] = enc.plan()
    print(json.dumps(row), flush=True)
