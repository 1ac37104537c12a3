#!/usr/bin/env python3
"""FASTECC_TRACE_PREPARE=1 python tools/trace_prepare.py: where the first fastecc_decode_prepare on the transform path spends its time."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import fastecc_amd as fe  # noqa: E402

p61 = len(sys.argv) > 1 and sys.argv[1] == "p61"   # the 64-bit field at (2^19, 2^18)
k = 1 << (18 if p61 else 19)
torch.zeros(1, device="cuda:0")
with fe.Encoder(2 * k, k, 4096, field=fe.FIELD_GF_P61_SQUARED if p61 else fe.FIELD_GF_FFF00001) as enc:
    rng = np.random.default_rng(1)
    for round_ in range(2):
        lost = rng.permutation(2 * k)[:2000]
        dp, pp = np.ones(k, np.uint8), np.ones(k, np.uint8)
        dp[lost[lost < k]] = 0
        pp[lost[lost >= k] - k] = 0
        t0 = time.perf_counter()
        enc.decode_prepare(dp, pp)
        print("prepare call %d: %.2f ms" % (round_, (time.perf_counter() - t0) * 1e3), flush=True)
