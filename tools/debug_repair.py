import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
import fastecc_amd as fe
k, S = 1 << 19, 1024
data = torch.randint(0, 0xFFF00001, (k * S,), dtype=torch.int64, device="cuda:0").to(torch.int32)
parity = torch.empty_like(data)
st = torch.cuda.current_stream().cuda_stream
with fe.Encoder(2 * k, k, 4096) as enc:
    enc.encode(data, parity, stream=st)
    rng = np.random.default_rng(1)
    lost = rng.permutation(2 * k)[:20000]
    dp, pp = np.ones(k, np.uint8), np.ones(k, np.uint8)
    dp[lost[lost < k]] = 0; pp[lost[lost >= k] - k] = 0
    enc.decode_prepare(dp, pp)
    w, wp = data.clone(), parity.clone()
    w.view(k, S)[torch.from_numpy(dp == 0).to("cuda:0")] = -1
    wp.view(k, S)[torch.from_numpy(pp == 0).to("cuda:0")] = -2
    enc.profile(True)
    enc.repair(w, wp, stream=st)
    torch.cuda.synchronize()
    print("ok", bool((w == data).all()), bool((wp == parity).all()))
    for name, v in enc.profile_read().items(): print(name, v)
