#!/usr/bin/env python3
"""The 64-bit field's (2k,k) direct path against its transform path at BASELINE configs[4]'s geometry (k = 2^19 x 64 KB): decode with e data blocks lost.
One JSON line per e."""
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
k, bb = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 19), 65536
data = random_stripe_p61(k * (bb // 8), dev, seed=0x619)
parity = torch.empty_like(data)
with fe.Encoder(2 * k, k, bb, field=fe.FIELD_GF_P61_SQUARED) as enc:
    enc.encode(data, parity)
    rng = np.random.default_rng(619)
    for e in (1, 4, 8, 16, 24, 32):
        row = {"lost_data_blocks": e}
        for name, dmax in (("direct_ms", 32), ("transform_ms", 0)):
            enc.set_option("decode_direct_max", dmax)
            dp, pp = np.ones(k, np.uint8), np.ones(k, np.uint8)
            lost = rng.permutation(k)[:e]
            dp[lost] = 0
            di = torch.from_numpy(np.sort(lost)).to(dev)
            dv = data.view(k, -1)
            saved = dv[di].clone()
            enc.decode_prepare(dp, pp)
            dv[di] = -1
            enc.decode(data, parity)
            torch.cuda.synchronize()
            ok = bool(torch.equal(dv[di], saved))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(2):
                enc.decode(data, parity)
            e1.record()
            torch.cuda.synchronize()
            row[name] = round(e0.elapsed_time(e1) / 2, 3)
            row[name.replace("_ms", "_restored")] = ok
        print(json.dumps(row), flush=True)
