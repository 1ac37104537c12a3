#!/usr/bin/env python3
"""Minimal driver for the profilers: direct (matrix-core) encodes of k + E codes at the headline size, E from argv (default 64,128).  No timing."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import fastecc_amd as fe  # noqa: E402

k, S = 1 << 19, 1024
stream = torch.cuda.current_stream().cuda_stream
data = torch.randint(0, 0xFFF00001, (k * S,), dtype=torch.int64, device="cuda:0").to(torch.int32)
for e in [int(a) for a in (sys.argv[1] if len(sys.argv) > 1 else "64,128").split(",")]:
    with fe.Encoder(k + e, k, 4 * S) as enc:
        enc.set_option("encode_direct_max", 256)
        enc.set_option("direct_kernel", 2)
        out = torch.empty(e * S, dtype=torch.int32, device="cuda:0")
        for _ in range(3):
            enc.encode(data, out, stream=stream)
torch.cuda.synchronize()
