#!/usr/bin/env python3
"""Minimal driver for the profilers: decode and repair of a 2 % loss pattern of the (2^20,2^19) x 4 KB code (the transform path). usage: [reps] [decode: skip the repairs]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import fastecc_amd as fe  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
k, S = 1 << 19, 1024
stream = torch.cuda.current_stream().cuda_stream
data = torch.randint(0, 0xFFF00001, (k * S,), dtype=torch.int64, device="cuda:0").to(torch.int32)
parity = torch.empty_like(data)
with fe.Encoder(2 * k, k, 4 * S) as enc:
    enc.encode(data, parity, stream=stream)
    lost = np.random.default_rng(2).permutation(2 * k)[: (2 * k) // 50]
    dp, pp = np.ones(k, np.uint8), np.ones(k, np.uint8)
    dp[lost[lost < k]] = 0
    pp[lost[lost >= k] - k] = 0
    enc.decode_prepare(dp, pp)
    for _ in range(reps):
        enc.decode(data, parity, stream=stream)
    for _ in range(0 if len(sys.argv) > 2 and sys.argv[2] == "decode" else reps):
        enc.repair(data, parity, stream=stream)
torch.cuda.synchronize()
