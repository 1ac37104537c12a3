#!/usr/bin/env python3
"""Encode time of the headline code with narrow blocks (512 B ... 4 KB): what one column slab of a sharded stripe costs on its GPU."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import fastecc_amd as fe  # noqa: E402

P = 0xFFF00001
k = 1 << 19
out = {}
for bb in (512, 256, 1024, 4096):
    S = bb // 4
    d = torch.randint(0, P, (k * S,), dtype=torch.int64, device="cuda:0").to(torch.int32)
    p = torch.empty_like(d)
    with fe.Encoder(2 * k, k, bb) as enc:
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(3): enc.encode(d, p, stream=st)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): enc.encode(d, p, stream=st)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 50 * 1e3
        out[bb] = {"ms": round(ms, 4), "GBps": round(2.0 * k * bb / ms / 1e6, 1), "plan": enc.plan()}
print(json.dumps(out))
