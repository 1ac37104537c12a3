#!/usr/bin/env python3
"""Other code shapes at k = 2^19 x 4 KB under the classic split (plan 3100: dif9 / mid10 / dit9) and the shorter MID (3090): fewer parity
blocks (fold), n = 4k (cosets), zero extension.  ms per encode (10 calls), same parity checked."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import fastecc_amd as fe  # noqa: E402

P = 0xFFF00001
S = 1024
k0 = 1 << 19
for name, k, m in (("fewer_parity_k/2", k0, k0 // 2), ("fewer_parity_k/8", k0, k0 // 8), ("n=4k", k0 // 2, 3 * (k0 // 2)), ("zero_extended_400000+100000", 400000, 100000),
                   ("zero_extended_500000+500000", 500000, 500000)):
    data = torch.randint(0, P, (k * S,), dtype=torch.int64, device="cuda:0").to(torch.int32)
    parity = torch.empty(m * S, dtype=torch.int32, device="cuda:0")
    row = {"code": name, "k": k, "m": m}
    with fe.Encoder(k + m, k, 4 * S) as enc:
        stream = torch.cuda.current_stream().cuda_stream
        ref = None
        for plan in (0, 3100, 3090, 3100, 3090):
            try:
                enc.set_plan(plan)
            except fe.FastEccError:
                row.setdefault(str(plan), []).append("unsupported")
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
            row.setdefault(str(plan), []).append([round(ms, 4), enc.plan(), bool(torch.equal(parity, ref))])
    print(json.dumps(row), flush=True)
