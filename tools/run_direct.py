#!/usr/bin/env python3
"""Minimal driver for the profilers: the direct paths at the headline size — repair of 1 + 1 and 4 + 4 lost blocks of the (2^20,2^19) code,
encode of a (2^19 + 4) code, and the mixed-radix encode of 3 * 2^17 blocks (fused odd-radix tiles).  No timing."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import fastecc_amd as fe  # noqa: E402

k, S = 1 << 19, 1024
stream = torch.cuda.current_stream().cuda_stream
data = torch.randint(0, 0xFFF00001, (k * S,), dtype=torch.int64, device="cuda:0").to(torch.int32)
parity = torch.empty_like(data)
with fe.Encoder(2 * k, k, 4 * S) as enc:
    enc.encode(data, parity, stream=stream)
    for ld, lp in ((1, 1), (4, 4)):
        dp, pp = np.ones(k, np.uint8), np.ones(k, np.uint8)
        dp[np.arange(ld) * 1000 + 7] = 0
        pp[np.arange(lp) * 3000 + 11] = 0
        enc.decode_prepare(dp, pp)
        for _ in range(2):
            enc.repair(data, parity, stream=stream)
with fe.Encoder(k + 4, k, 4 * S) as enc:
    small = torch.empty(4 * S, dtype=torch.int32, device="cuda:0")
    for _ in range(2):
        enc.encode(data, small, stream=stream)
km = 3 << 17
with fe.Encoder(2 * km, km, 4 * S, flags=fe.CODE_MIXED_RADIX) as enc:
    for _ in range(2):
        enc.encode(data[: km * S], parity[: km * S], stream=stream)
torch.cuda.synchronize()
