#!/usr/bin/env python3
"""What the host link of this box gives the FASTECC_MEM_HOST_PINNED pipeline: contiguous 2 GiB copies up / down / both, the strided 2-D
copies the pipeline issues per column slab (hipMemcpy2DAsync, rows of 4096 / H bytes), and the pipeline itself for H = 2 .. 32."""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import fastecc_amd as fe  # noqa: E402

hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
hip.hipMemcpy2DAsync.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
H2D, D2H = 1, 2
N, S = 1 << 19, 1024
dev = torch.randint(0, 0xFFF00001, (N * S,), dtype=torch.int64, device="cuda:0").to(torch.int32)
dev2 = torch.empty_like(dev)
hx = torch.empty(N * S, dtype=torch.int32).pin_memory()
hp = torch.empty(N * S, dtype=torch.int32).pin_memory()
hx.copy_(dev)
s_up, s_dn = torch.cuda.Stream(), torch.cuda.Stream()
nbytes = float(N * S * 4)


def wall(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def up2d(width, col0=0):
    hip.hipMemcpy2DAsync(dev.data_ptr() + col0, S * 4, hx.data_ptr() + col0, S * 4, width, N, H2D, s_up.cuda_stream)


def dn2d(width, col0=0):
    hip.hipMemcpy2DAsync(hp.data_ptr() + col0, S * 4, dev2.data_ptr() + col0, S * 4, width, N, D2H, s_dn.cuda_stream)


out = {}
ms = wall(lambda: up2d(S * 4))
out["contiguous_h2d_GBps"] = round(nbytes / ms / 1e6, 1)
ms = wall(lambda: dn2d(S * 4))
out["contiguous_d2h_GBps"] = round(nbytes / ms / 1e6, 1)
ms = wall(lambda: (up2d(S * 4), dn2d(S * 4)))
out["contiguous_both_GBps"] = round(2 * nbytes / ms / 1e6, 1)
for H in (2, 4, 8, 16, 32):
    w = S * 4 // H
    def all_up():
        for h in range(H):
            up2d(w, h * w)
    def all_dn():
        for h in range(H):
            dn2d(w, h * w)
    out["strided_%dB_rows" % w] = {"h2d_GBps": round(nbytes / wall(all_up) / 1e6, 1), "d2h_GBps": round(nbytes / wall(all_dn) / 1e6, 1),
                                    "both_GBps": round(2 * nbytes / wall(lambda: (all_up(), all_dn())) / 1e6, 1)}
print(json.dumps(out), flush=True)
with fe.Encoder(2 * N, N, 4 * S) as enc:
    st = torch.cuda.current_stream().cuda_stream
    res = {}
    for H in (2, 4, 8, 16, 32):
        enc.set_option("host_slabs", H)
        def once():
            enc.encode(hx.data_ptr(), hp.data_ptr(), stream=st, mem=fe.MEM_HOST_PINNED)
            torch.cuda.synchronize()
        ms = wall(once, 4)
        res["host_slabs_%d" % H] = {"ms": round(ms, 2), "GBps": round(2 * nbytes / ms / 1e6, 1)}
    print(json.dumps(res), flush=True)
