#!/usr/bin/env python3
"""fastecc_encode_blocks — the reference's T** form (RS.cpp:25-33: N pointers to blocks, encoded in place) — at the headline size, host memory."""
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import fastecc_amd as fe  # noqa: E402

log2k = int(sys.argv[1]) if len(sys.argv) > 1 else 19
N, S = 1 << log2k, 1024
x = (np.arange(N * S, dtype=np.uint64) % 0xFFF00001).astype(np.uint32).reshape(N, S)
want = np.empty_like(x)
with fe.Encoder(2 * N, N, 4 * S) as enc:
    if len(sys.argv) > 2:
        enc.set_option("stage_threads", int(sys.argv[2]))
    enc.encode_host(x, want)
    ts = []
    for rep in range(3):
        out = np.empty_like(x)
        t0 = time.perf_counter()
        enc.encode_host(x, out)
        ts.append((time.perf_counter() - t0) * 1e3)
    print(json.dumps({"call": "fastecc_encode(FASTECC_MEM_HOST)", "stage_threads": int(sys.argv[2]) if len(sys.argv) > 2 else 0, "ms": [round(t, 1) for t in ts]}), flush=True)
    for layout in ("one_buffer_in_order", "one_buffer_shuffled"):
        ts = []
        for rep in range(3):
            buf = x.copy()
            order = np.arange(N)
            if layout.endswith("shuffled"):
                np.random.default_rng(1).shuffle(order)  # block i lives at slot order[i]
                buf[order] = x
            ptrs = (buf.ctypes.data + order.astype(np.uint64) * np.uint64(4 * S)).tolist()
            arr = (ctypes.c_void_p * N)(*ptrs)  # (building this table is the caller's business: not timed)
            t0 = time.perf_counter()
            rc = fe.lib().fastecc_encode_blocks(enc._h, arr)
            ts.append((time.perf_counter() - t0) * 1e3)
            assert rc == 0, rc
            ok = bool(np.array_equal(buf[order], want))
        print(json.dumps({"k": N, "block_bytes": 4 * S, "layout": layout, "ms": [round(t, 1) for t in ts], "GBps": round(2.0 * x.nbytes / min(ts) / 1e6, 1), "parity_ok": ok}), flush=True)
