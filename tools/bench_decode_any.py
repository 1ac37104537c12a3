#!/usr/bin/env python3
"""Few-loss repair of codes that are not the (2k,k) power-of-two layout: zero-extended (400000 + 100000), fewer parity blocks
(2^19 + 2^16), mixed radix (3 * 2^17 + 3 * 2^17), 4 KB blocks; the direct interpolation path against the transform path
(decode_direct_max = 0).  One JSON line per code."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import fastecc_amd as fe  # noqa: E402

S = 1024
stream = torch.cuda.current_stream().cuda_stream
for name, k, m, flags in (("zero_extended", 400000, 100000, 0), ("fewer_parity", 1 << 19, 1 << 16, 0), ("mixed_radix", 3 << 17, 3 << 17, fe.CODE_MIXED_RADIX)):
    data = torch.randint(0, 0xFFF00001, (k * S,), dtype=torch.int64, device="cuda:0").to(torch.int32)
    parity = torch.empty(m * S, dtype=torch.int32, device="cuda:0")
    row = {"code": name, "k": k, "m": m, "block_bytes": 4 * S, "cases": []}
    with fe.Encoder(k + m, k, 4 * S, flags=flags) as enc:
        enc.encode(data, parity, stream=stream)
        for lost_data, lost_parity in ((1, 0), (1, 1), (4, 4), (8, 8)):
            rng = np.random.default_rng(lost_data * 17 + lost_parity)
            dp, pp = np.ones(k, np.uint8), np.ones(m, np.uint8)
            dp[rng.permutation(k)[:lost_data]] = 0
            pp[rng.permutation(m)[:lost_parity]] = 0
            case = {"lost_data": lost_data, "lost_parity": lost_parity}
            for direct_max in (16, 0):
                enc.set_option("decode_direct_max", direct_max)
                enc.decode_prepare(dp, pp)  # first call of a path may build its tables
                t0 = time.perf_counter()
                enc.decode_prepare(dp, pp)
                prep = (time.perf_counter() - t0) * 1e3
                wd, wp = data.clone(), parity.clone()
                wd.view(k, S)[torch.from_numpy(dp == 0).to("cuda:0")] = -1
                wp.view(m, S)[torch.from_numpy(pp == 0).to("cuda:0")] = -2
                enc.repair(wd, wp, stream=stream)
                ok = bool(torch.equal(wd, data)) and bool(torch.equal(wp, parity))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(10):
                    enc.repair(wd, wp, stream=stream)
                e1.record()
                torch.cuda.synchronize()
                case["direct" if direct_max else "transform"] = {"prepare_ms": round(prep, 2), "repair_ms": round(e0.elapsed_time(e1) / 10, 3), "ok": ok}
            row["cases"].append(case)
        enc.set_option("decode_direct_max", 16)
    print(json.dumps(row), flush=True)
    del data, parity
