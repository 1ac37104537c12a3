#!/usr/bin/env python3
"""Time direct.hip at the headline code, (n,k) = (2^20, 2^19) x 4 KB, HBM-resident: decode of E lost DATA blocks (one pass over
k + E blocks) and encode of k + m codes (one pass over k blocks), by the VALU kernel (1) and the MFMA kernel (2), beside the
transform path.  HIP events on the launch stream; every timed result is checked.  One JSON line per case."""
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


def timed(fn, steps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


def main():
    log2k = int(sys.argv[1]) if len(sys.argv) > 1 else 19
    counts = [int(a) for a in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 4, 8, 16, 17, 32, 64, 128, 256]
    N, S, steps = 1 << log2k, 1024, 10
    g = torch.Generator(device="cuda:0").manual_seed(11)
    data = torch.randint(0, P, (N * S,), dtype=torch.int64, device="cuda:0", generator=g).to(torch.int32)
    parity = torch.empty_like(data)
    stream = torch.cuda.current_stream().cuda_stream
    with fastecc_amd.Encoder(2 * N, N, 4096) as enc:
        enc.encode(data, parity, stream=stream)
        enc.set_option("decode_direct_max", 256)
        for e in counts:
            rng = np.random.default_rng(e)
            dp, pp = np.ones(N, np.uint8), np.ones(N, np.uint8)
            dp[rng.permutation(N)[:e]] = 0
            row = {"case": "decode", "lost_data_blocks": e}
            for name, kernel, dmax in (("valu", 1, 256), ("mfma", 2, 256), ("transform", 0, 0)):
                if name == "transform" and e not in (counts[0], counts[-1]):
                    continue
                enc.set_option("direct_kernel", kernel)
                enc.set_option("decode_direct_max", dmax)
                enc.decode_prepare(dp, pp)
                t0 = time.perf_counter()
                enc.decode_prepare(dp, pp)
                prep = (time.perf_counter() - t0) * 1e3
                work = data.clone()
                work.view(N, S)[torch.from_numpy(dp == 0).to("cuda:0")] = -1
                enc.decode(work, parity, stream=stream)
                ok = bool((work == data).all())
                ms = timed(lambda: enc.decode(work, parity, stream=stream), steps)
                row[name] = {"prepare_ms": round(prep, 2), "decode_ms": round(ms, 3), "ok": ok, "data_read_TBps": round(N * 4096 / (ms * 1e-3) / 1e12, 2)}
            print(json.dumps(row), flush=True)
    for m in counts:
        row = {"case": "encode", "parity_blocks": m}
        with fastecc_amd.Encoder(N + m, N, 4096) as enc:
            out = torch.empty(m * S, dtype=torch.int32, device="cuda:0")
            ref = None
            for name, kernel, dmax in (("pipeline", 0, 0), ("valu", 1, 256), ("mfma", 2, 256)):
                if name == "valu" and (m > 64 or os.environ.get("FASTECC_BENCH_DIRECT_FAST")):
                    continue
                enc.set_option("direct_kernel", kernel)
                enc.set_option("encode_direct_max", dmax)
                enc.encode(data, out, stream=stream)
                torch.cuda.synchronize()
                if ref is None:
                    ref = out.clone()
                ok = bool((out == ref).all())
                ms = timed(lambda: enc.encode(data, out, stream=stream), steps)
                row[name] = {"ms": round(ms, 3), "same_as_pipeline": ok, "data_read_TBps": round(N * 4096 / (ms * 1e-3) / 1e12, 2)}
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
