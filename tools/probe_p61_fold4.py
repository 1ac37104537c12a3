import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import fastecc_amd as fe
P61 = (1 << 61) - 1
for logn in range(5, 15):
    for elems in (2, 66):
        N = 1 << logn; e = 2; rows = 3 * N
        g = torch.Generator(device="cuda:0"); g.manual_seed(logn)
        x = torch.randint(0, P61, (N * 2 * elems,), dtype=torch.int64, device="cuda:0", generator=g)
        par = torch.empty(rows * 2 * elems, dtype=torch.int64, device="cuda:0")
        with fe.Encoder(N << e, N, 16 * elems, field=fe.FIELD_GF_P61_SQUARED) as enc:
            enc.encode(x, par)
            rng = np.random.default_rng(logn)
            lost = rng.permutation(4 * N)[: 3 * N]
            dp, pp = np.ones(N, np.uint8), np.ones(rows, np.uint8)
            dp[lost[lost < N]] = 0; pp[lost[lost >= N] - N] = 0
            d = x.clone(); d.view(N, 2 * elems)[torch.from_numpy(dp == 0).to("cuda:0")] = -1
            enc.decode_prepare(dp, pp)
            enc.profile(True); enc.profile_reset()
            enc.decode(d, par); torch.cuda.synchronize()
            prof = enc.profile_read(); enc.profile(False)
            print(logn, elems, int((dp == 0).sum()), "ok" if torch.equal(d, x) else "WRONG", sorted(prof), flush=True)
