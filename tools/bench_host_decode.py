#!/usr/bin/env python3
"""fastecc_decode / fastecc_repair of HOST-resident stripes (FASTECC_MEM_HOST, pageable memory) at the headline code, wall clock: the stripes
are staged through HBM — of the parity stripe only the block groups the decoder reads — and with few lost blocks only the rebuilt ones come back.
One JSON line."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import fastecc_amd as fe  # noqa: E402

P = 0xFFF00001
k, S = 1 << 19, 1024
g = torch.Generator(device="cuda:0").manual_seed(1)
data = torch.randint(0, P, (k * S,), dtype=torch.int64, device="cuda:0", generator=g).to(torch.int32)
parity = torch.empty_like(data)
out = {"workload": "(n,k)=(2^20,2^19), 4096 B blocks, stripes in pageable host memory (2 + 2 GiB)", "cases": []}
with fe.Encoder(2 * k, k, 4 * S) as enc:
    enc.encode(data, parity)
    torch.cuda.synchronize()
    x = data.cpu().numpy().view(np.uint32).reshape(k, S)
    par = parity.cpu().numpy().view(np.uint32).reshape(k, S)
    rng = np.random.default_rng(3)
    for count in (16, 2 * k // 50, 2 * k // 4):
        lost = rng.permutation(2 * k)[:count]
        dp, pp = np.ones(k, np.uint8), np.ones(k, np.uint8)
        dp[lost[lost < k]] = 0
        pp[lost[lost >= k] - k] = 0
        enc.decode_prepare(dp, pp)
        rec = {"lost_blocks": int(count)}
        for name in ("decode", "repair"):
            best = 1e9
            for _ in range(3):
                hd, hq = x.copy(), par.copy()
                hd[dp == 0] = 7
                hq[pp == 0] = 9
                t0 = time.perf_counter()
                (enc.decode if name == "decode" else enc.repair)(hd, hq, mem=fe.MEM_HOST)
                best = min(best, (time.perf_counter() - t0) * 1e3)
                ok = bool(np.array_equal(hd, x)) and (name == "decode" or bool(np.array_equal(hq, par)))
            rec[name + "_ms"] = round(best, 1)
            rec[name + "_ok"] = ok
        out["cases"].append(rec)
print(json.dumps(out))
