#!/usr/bin/env python3
"""fastecc_decode_prepare (wall clock, 8 calls) and fastecc_repair (HIP events) for the few-loss patterns of bench.py's other_paths at k = 2^19 x 4 KB;
run it again with FASTECC_HIP_LIB=<another build of the library> for an A/B on one box."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fastecc_amd as fe
k = 1 << 19
torch.zeros(1, device="cuda:0")
rng = np.random.default_rng(7)
with fe.Encoder(2 * k, k, 4096) as enc:
    d = torch.randint(0, 1 << 30, (k * 1024,), dtype=torch.int32, device="cuda:0"); q = torch.empty_like(d)
    enc.encode(d, q); torch.cuda.synchronize()
    for name, lost in (("1+1", np.array([k // 3, k + k // 7])), ("8+8", np.r_[rng.permutation(k)[:8], k + rng.permutation(k)[:8]]), ("16+0", rng.permutation(k)[:16]),
                       ("64+0", rng.permutation(k)[:64]), ("128+128", np.r_[rng.permutation(k)[:128], k + rng.permutation(k)[:128]])):
        dp, pp = np.ones(k, np.uint8), np.ones(k, np.uint8)
        dp[lost[lost < k]] = 0; pp[lost[lost >= k] - k] = 0
        enc.decode_prepare(dp, pp)
        ts = []
        for _ in range(8):
            t0 = time.perf_counter(); enc.decode_prepare(dp, pp); ts.append((time.perf_counter() - t0) * 1e3)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        enc.repair(d, q); torch.cuda.synchronize(); e0.record()
        for _ in range(5): enc.repair(d, q)
        e1.record(); torch.cuda.synchronize()
        print(name, "prepare min %.3f med %.3f ms" % (min(ts), sorted(ts)[4]), "repair %.3f ms" % (e0.elapsed_time(e1) / 5), flush=True)
