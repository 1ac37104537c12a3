#!/usr/bin/env python3
"""A (k + m, k) code encoded by the direct path (one read of the data), a few times — for rocprofv3: python tools/run_direct_encode.py [m] [log2k]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import fastecc_amd as fe  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 64
k = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 19)
d = torch.randint(0, 1 << 30, (k * 1024,), dtype=torch.int32, device="cuda:0")
q = torch.empty(m * 1024, dtype=torch.int32, device="cuda:0")
with fe.Encoder(k + m, k, 4096) as enc:
    for _ in range(4):
        enc.encode(d, q)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        enc.encode(d, q)
    e1.record()
    torch.cuda.synchronize()
    print("encode ms", e0.elapsed_time(e1) / 10)
