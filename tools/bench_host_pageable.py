#!/usr/bin/env python3
"""fastecc_encode(FASTECC_MEM_HOST) on pageable buffers at the headline size: the column-slab pipeline through the staging rings against the
upload / encode / download sequence, per slab count."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import fastecc_amd as fe  # noqa: E402

N, S = 1 << 19, 1024
x = (np.arange(N * S, dtype=np.uint64) % 0xFFF00001).astype(np.uint32).reshape(N, S)
out = np.empty_like(x)
ref = None
with fe.Encoder(2 * N, N, 4 * S) as enc:
    for pipeline, slabs in ((0, 8), (1, 2), (1, 4), (1, 8), (1, 16)):
        enc.set_option("host_pipeline", pipeline)
        enc.set_option("host_slabs", slabs)
        enc.encode_host(x, out)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            enc.encode_host(x, out)
            ts.append((time.perf_counter() - t0) * 1e3)
        if ref is None:
            ref = out.copy()
        print(json.dumps({"host_pipeline": pipeline, "host_slabs": slabs, "ms": [round(t, 1) for t in ts], "GBps": round(2.0 * x.nbytes / min(ts) / 1e6, 1),
                          "same_parity": bool(np.array_equal(out, ref))}), flush=True)
