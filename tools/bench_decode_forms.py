#!/usr/bin/env python3
"""The transform decoder's forms at (2^20, 2^19) x 4 KB, 2 % of the codeword lost: the split transform in its small form (default), in the
block-group form (decode_split = 2) and the unsplit 2k-point transform (decode_split = 0: 1024-block tiles of two address windows), decode
and repair, HIP events over 10 calls, with the per-kernel averages."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import fastecc_amd as fe  # noqa: E402

k, S = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 19), 1024
stream = torch.cuda.current_stream().cuda_stream
data = torch.randint(0, 0xFFF00001, (k * S,), dtype=torch.int64, device="cuda:0").to(torch.int32)
parity = torch.empty_like(data)


def event_ms(fn, reps=10):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


with fe.Encoder(2 * k, k, 4 * S) as enc:
    enc.encode(data, parity, stream=stream)
    lost = np.random.default_rng(2).permutation(2 * k)[: (2 * k) // 50]
    dp, pp = np.ones(k, np.uint8), np.ones(k, np.uint8)
    dp[lost[lost < k]] = 0
    pp[lost[lost >= k] - k] = 0
    saved = data.clone()
    for form, split in (("split_small_form", 1), ("split_block_groups", 2), ("unsplit_2k_point_transform", 0)):
        enc.set_option("decode_split", split)
        enc.decode_prepare(dp, pp)
        data.view(k, S)[torch.from_numpy(np.flatnonzero(dp == 0)).to("cuda:0")] = -1
        enc.decode(data, parity, stream=stream)
        torch.cuda.synchronize()
        row = {"form": form, "log2k": int(np.log2(k)), "restored": bool(torch.equal(data, saved)), "decode_ms": round(event_ms(lambda: enc.decode(data, parity, stream=stream)), 3),
               "repair_ms": round(event_ms(lambda: enc.repair(data, parity, stream=stream)), 3)}
        enc.profile(True)
        enc.profile_reset()
        for _ in range(3):
            enc.decode(data, parity, stream=stream)
        row["decode_kernel_ms"] = {kn: round(v[0] / v[1], 4) for kn, v in enc.profile_read().items()}
        enc.profile(False)
        print(json.dumps(row), flush=True)
