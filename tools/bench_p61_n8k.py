#!/usr/bin/env python3
"""The 64-bit field's n = 8k code at (2^19, 2^16) x 64 KB: encode (4 GiB of data -> 28 GiB of parity), decode with 2 % of the data blocks lost
(the folded transform: 2k outputs instead of 8k) and with 3 blocks lost (the inner (2k,k) code's direct path).  One JSON line."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import fastecc_amd as fe  # noqa: E402
from bench_common import random_stripe_p61  # noqa: E402

dev = torch.device("cuda", 0)
k, bb, e = 1 << 16, 65536, 3
data = random_stripe_p61(k * (bb // 8), dev, seed=0x618)
parity = torch.empty(7 * data.numel(), dtype=data.dtype, device=dev)


def timed(fn, reps):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


out = {}
with fe.Encoder(k << e, k, bb, field=fe.FIELD_GF_P61_SQUARED) as enc:
    out["encode_ms"] = round(timed(lambda: enc.encode(data, parity), 3), 3)
    rng = np.random.default_rng(618)
    for name, lost_d, lost_p in (("decode_2_percent_of_the_data_lost", rng.permutation(k)[: k // 50], []), ("decode_3_data_1_parity_lost", [5, k // 3, k - 1], [7])):
        dp, pp = np.ones(k, np.uint8), np.ones(7 * k, np.uint8)
        dp[lost_d] = 0
        pp[lost_p] = 0
        di = torch.from_numpy(np.flatnonzero(dp == 0)).to(dev)
        dv = data.view(k, -1)
        saved = dv[di].clone()
        enc.decode_prepare(dp, pp)
        dv[di] = -1
        enc.profile(True)
        enc.profile_reset()
        enc.decode(data, parity)
        torch.cuda.synchronize()
        kernels = sorted(enc.profile_read())
        enc.profile(False)
        ok = bool(torch.equal(dv[di], saved))
        out[name] = {"ms": round(timed(lambda: enc.decode(data, parity), 2), 3), "restored": ok, "kernels": kernels}
print(json.dumps(out))
