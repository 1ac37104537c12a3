#!/usr/bin/env python3
"""Time the erasure decoder at the headline code: (n,k) = (2^20, 2^19), 4 KB blocks, HBM-resident codeword.
For each loss rate: pattern preparation (fastecc_decode_prepare: ms, once per erasure pattern; device product tree) and the per-stripe decode on the GPU
(HIP events on the stream the kernels run on).  GB/s uses the codeword bytes a decode reads (data + parity = 4 GiB).
One JSON line."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import fastecc_amd  # noqa: E402

P = 0xFFF00001


def main():
    log2k = int(sys.argv[1]) if len(sys.argv) > 1 else 19
    p61 = len(sys.argv) > 2 and sys.argv[2] == "p61"   # GF((2^61-1)^2): S counts 64-bit words, 4096-byte blocks as well
    N, S, steps = 1 << log2k, (512 if p61 else 1024), 10
    g = torch.Generator(device="cuda:0").manual_seed(11)
    if p61:
        data = torch.randint(0, (1 << 61) - 1, (N * S,), dtype=torch.int64, device="cuda:0", generator=g)
    else:
        data = torch.randint(0, P, (N * S,), dtype=torch.int64, device="cuda:0", generator=g).to(torch.int32)
    parity = torch.empty_like(data)
    out = {"workload": "(n,k)=(2^%d,2^%d), 4096 B blocks, %s, random erasures over the whole codeword"
                       % (log2k + 1, log2k, "GF((2^61-1)^2)" if p61 else "GF(0xFFF00001)"), "cases": []}
    stream = torch.cuda.current_stream().cuda_stream
    with fastecc_amd.Encoder(2 * N, N, 4096, field=fastecc_amd.FIELD_GF_P61_SQUARED if p61 else fastecc_amd.FIELD_GF_FFF00001) as enc:
        enc.encode(data, parity, stream=stream)
        # a few lost blocks (the direct path of the 32-bit field's (2k,k) codes), then loss rates (locator tree + transform)
        for frac in (1, 2, 4, 8, 16, 0.001, 0.02, 0.25, 0.5):
            rng = np.random.default_rng(int(frac * 1000))
            lost = rng.permutation(2 * N)[: (frac if isinstance(frac, int) else max(1, int(2 * N * frac)))]
            if isinstance(frac, int):
                lost[0] = 5 * 2  # at least one data block among them
            dp, pp = np.ones(N, np.uint8), np.ones(N, np.uint8)
            dp[lost[lost < N]] = 0
            pp[lost[lost >= N] - N] = 0
            t0 = time.perf_counter()
            enc.decode_prepare(dp, pp)
            first_ms = (time.perf_counter() - t0) * 1e3  # the very first call also builds the decoder's contexts and tables
            t0 = time.perf_counter()
            enc.decode_prepare(dp, pp)
            prep_ms = (time.perf_counter() - t0) * 1e3
            work = data.clone()
            work.view(N, S)[torch.from_numpy(dp == 0).to("cuda:0")] = -1
            wpar = parity.clone()
            wpar.view(N, S)[torch.from_numpy(pp == 0).to("cuda:0")] = -2
            enc.repair(work, wpar, stream=stream)  # warm-up (allocations), and the check of what is timed below
            assert bool((work == data).all()) and bool((wpar == parity).all())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(steps):
                enc.decode(work, parity, stream=stream)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / steps
            e0.record()
            for _ in range(steps):
                enc.repair(work, wpar, stream=stream)
            e1.record()
            torch.cuda.synchronize()
            repair_ms = e0.elapsed_time(e1) / steps
            out["cases"].append({"lost": frac, "erased_blocks": int(np.unique(lost).size), "repair_ms": round(repair_ms, 3), "erased_data_blocks": int((dp == 0).sum()),
                                 "prepare_ms": round(prep_ms, 2), "prepare_first_call_ms": round(first_ms, 1), "decode_ms": round(ms, 3),
                                 "codeword_GBps": round(2.0 * N * 4096 / (ms * 1e-3) / 1e9, 1)})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
