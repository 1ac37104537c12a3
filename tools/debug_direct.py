#!/usr/bin/env python3
"""Debug aid: small direct-path round trips, printing which case / kernel / rows differ."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fastecc_amd as fe
P = 0xFFF00001
def to_dev(a): return torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to("cuda:0")
def to_host(t, shape): return t.cpu().numpy().view(np.uint32).reshape(shape)
cases = [(int(a.split("x")[0]), int(a.split("x")[1])) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [(96, 64), (128, 64), (32, 64), (100, 64), (96, 66)]
for N, S in cases:
    rng = np.random.default_rng(N + S)
    x = rng.integers(0, P, size=(N, S), dtype=np.uint64).astype(np.uint32)
    with fe.Encoder(2 * N, N, 4 * S) as enc:
        pd = torch.empty(N * S, dtype=torch.int32, device="cuda:0")
        enc.encode(to_dev(x), pd)
        torch.cuda.synchronize()
        par = to_host(pd, (N, S)).copy()
        for e in (1, 2, 5, 8, 15, 16, 17, 24, 32, 33, 48, 64, 65):
            if e > N: continue
            lost = np.unique(np.r_[int(rng.integers(0, N)), rng.permutation(2 * N)[: e - 1]])
            dp, pp = np.ones(N, np.uint8), np.ones(N, np.uint8)
            dp[lost[lost < N]] = 0; pp[lost[lost >= N] - N] = 0
            bad_x, bad_p = x.copy(), par.copy(); bad_x[dp == 0] = 0xA5A5A5A5; bad_p[pp == 0] = 0x5A5A5A5A
            for name, kernel in (("valu", 1), ("mfma", 2)):
                enc.set_option("decode_direct_max", 256); enc.set_option("direct_kernel", kernel)
                enc.decode_prepare(dp, pp)
                d, q = to_dev(bad_x), to_dev(bad_p)
                enc.repair(d, q)
                torch.cuda.synchronize()
                gd, gq = to_host(d, (N, S)), to_host(q, (N, S))
                bd, bq = np.flatnonzero((gd != x).any(axis=1)), np.flatnonzero((gq != par).any(axis=1))
                print(N, S, "e=%d" % e, name, "lost data", int((dp == 0).sum()), "parity", int((pp == 0).sum()), "OK" if len(bd) + len(bq) == 0 else
                      "BAD data rows %s parity rows %s (lost data rows %s)" % (bd[:8], bq[:8], np.flatnonzero(dp == 0)[:8]), flush=True)
