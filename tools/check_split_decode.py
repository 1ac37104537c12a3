#!/usr/bin/env python3
"""fastecc_decode through the split transform ("decode_split" = 1) against the 2k-point transform (= 0) and the original stripe:
python tools/check_split_decode.py [log2k] [words per block] [parity blocks (default k)]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import fastecc_amd as fe  # noqa: E402

log2k = int(sys.argv[1]) if len(sys.argv) > 1 else 18
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
k = 1 << log2k
m = int(sys.argv[3]) if len(sys.argv) > 3 else k
stream = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device="cuda:0").manual_seed(3)
data = torch.randint(0, 0xFFF00001, (k * S,), dtype=torch.int64, device="cuda:0", generator=g).to(torch.int32)
parity = torch.empty(m * S, dtype=torch.int32, device="cuda:0")
ok = True
with fe.Encoder(k + m, k, 4 * S) as enc:
    enc.encode(data, parity, stream=stream)
    rng = np.random.default_rng(7)
    for frac in (0.0005, 0.02, 0.3, 0.5):
        lost = rng.permutation(k + m)[: min(m, max(300, int((k + m) * frac)))]
        dp, pp = np.ones(k, np.uint8), np.ones(m, np.uint8)
        dp[lost[lost < k]] = 0
        pp[lost[lost >= k] - k] = 0
        res = {}
        for split in (1, 0):
            enc.set_option("decode_split", split)
            enc.decode_prepare(dp, pp)
            work = data.clone()
            work.view(k, S)[torch.from_numpy(dp == 0).to("cuda:0")] = -1
            wpar = parity.clone()
            wpar.view(m, S)[torch.from_numpy(pp == 0).to("cuda:0")] = -2
            enc.decode(work, wpar, stream=stream)
            torch.cuda.synchronize()
            good = bool((work == data).all())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                enc.decode(work, wpar, stream=stream)
            e1.record()
            torch.cuda.synchronize()
            res[split] = (good, round(e0.elapsed_time(e1) / 5, 3))
            if os.environ.get("FASTECC_CHECK_PROFILE"):
                enc.profile(True)
                enc.profile_reset()
                for _ in range(5):
                    enc.decode(work, wpar, stream=stream)
                torch.cuda.synchronize()
                print("  split=%d" % split, {name: round(v[0] / v[1], 3) for name, v in enc.profile_read().items()}, flush=True)
                enc.profile(False)
            ok &= good
        print("lost %.4f: split %s, 2k-point %s" % (frac, res[1], res[0]), flush=True)
print("OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
