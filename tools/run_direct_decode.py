#!/usr/bin/env python3
"""Minimal driver for the profilers: decode of E lost data blocks of the (2^20,2^19) x 4 KB code by the direct path. usage: E [kernel] [reps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import fastecc_amd as fe  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 64
kernel = int(sys.argv[2]) if len(sys.argv) > 2 else 0
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
k, S = 1 << 19, 1024
stream = torch.cuda.current_stream().cuda_stream
data = torch.randint(0, 0xFFF00001, (k * S,), dtype=torch.int64, device="cuda:0").to(torch.int32)
parity = torch.empty_like(data)
with fe.Encoder(2 * k, k, 4 * S) as enc:
    enc.encode(data, parity, stream=stream)
    enc.set_option("decode_direct_max", 256)
    enc.set_option("direct_kernel", kernel)
    dp, pp = np.ones(k, np.uint8), np.ones(k, np.uint8)
    dp[np.random.default_rng(E).permutation(k)[:E]] = 0
    enc.decode_prepare(dp, pp)
    for _ in range(reps):
        enc.decode(data, parity, stream=stream)
torch.cuda.synchronize()
