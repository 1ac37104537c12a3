#!/usr/bin/env python3
"""fastecc_decode_prepare for a random pattern of the given loss fraction, several times (for rocprofv3 / FASTECC_TRACE_PREPARE):
python tools/run_prepare.py <fraction of the codeword lost> [calls] [p61]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import fastecc_amd as fe  # noqa: E402

frac = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 5
p61 = len(sys.argv) > 3 and sys.argv[3] == "p61"
warm = len(sys.argv) > 3 and sys.argv[3] == "warm"  # an encode first, as in bench.py's other_paths: the encoder's kernels are loaded before the first prepare
k = 1 << (18 if p61 else 19)
torch.zeros(1, device="cuda:0")
with fe.Encoder(2 * k, k, 4096, field=fe.FIELD_GF_P61_SQUARED if p61 else fe.FIELD_GF_FFF00001) as enc:
    rng = np.random.default_rng(1)
    if warm:
        d = torch.zeros(k * 1024, dtype=torch.int32, device="cuda:0")
        q = torch.empty_like(d)
        enc.encode(d, q)
        torch.cuda.synchronize()
    for call in range(calls):
        lost = rng.permutation(2 * k)[: max(1, int(2 * k * frac))]
        dp, pp = np.ones(k, np.uint8), np.ones(k, np.uint8)
        dp[lost[lost < k]] = 0
        pp[lost[lost >= k] - k] = 0
        t0 = time.perf_counter()
        enc.decode_prepare(dp, pp)
        print("prepare call %d: %.2f ms" % (call, (time.perf_counter() - t0) * 1e3), flush=True)
