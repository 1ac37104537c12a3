#!/usr/bin/env python3
"""End-to-end encode of a HOST-resident stripe at the headline size (PCIe-inclusive; never bench.py's `value`):
FASTECC_MEM_HOST (pageable memory, staged with copies) against FASTECC_MEM_HOST_PINNED (pinned memory, column slabs:
copy-engine upload, kernels and download of different slabs overlap).  One JSON line."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import fastecc_amd as fe  # noqa: E402

P = 0xFFF00001


def main():
    log2k = int(sys.argv[1]) if len(sys.argv) > 1 else 19
    N, S = 1 << log2k, 1024
    g = torch.Generator(device="cuda:0").manual_seed(1)
    dev = torch.randint(0, P, (N * S,), dtype=torch.int64, device="cuda:0", generator=g).to(torch.int32)
    want = torch.empty_like(dev)
    nbytes = 2.0 * N * S * 4
    out = {"workload": "(n,k)=(2^%d,2^%d), 4096 B blocks, stripe resident in host memory" % (log2k + 1, log2k)}
    with fe.Encoder(2 * N, N, 4 * S) as enc:
        enc.encode(dev, want)
        torch.cuda.synchronize()
        # pinned, kernels cross the link
        hdata = torch.empty(N * S, dtype=torch.int32).pin_memory()
        hpar = torch.empty(N * S, dtype=torch.int32).pin_memory()
        hdata.copy_(dev)
        stream = torch.cuda.current_stream().cuda_stream
        for slabs in (2, 4, 8):
            enc.set_option("host_slabs", slabs)
            hpar.zero_()
            best = None
            for it in range(4):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                enc.encode(hdata.data_ptr(), hpar.data_ptr(), stream=stream, mem=fe.MEM_HOST_PINNED)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                if it:
                    best = dt if best is None else min(best, dt)
            assert bool((hpar.cuda() == want).all())
            out["host_pinned_%d_slabs" % slabs] = {"ms": round(best * 1e3, 2), "GBps": round(nbytes / best / 1e9, 1)}
        # the same with explicit copies around a device encode (pinned memory, one stream): the no-overlap reference point
        d2 = torch.empty_like(dev)
        p2 = torch.empty_like(dev)
        best = None
        for it in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            d2.copy_(hdata, non_blocking=True)
            enc.encode(d2, p2, stream=stream)
            hpar.copy_(p2, non_blocking=True)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if it:
                best = dt if best is None else min(best, dt)
        out["pinned_copy_encode_copy"] = {"ms": round(best * 1e3, 2), "GBps": round(nbytes / best / 1e9, 1)}
        # pageable memory through FASTECC_MEM_HOST (what RS.cpp's malloc'ed buffers get)
        x = hdata.numpy().view(np.uint32).reshape(N, S).copy()
        y = np.empty_like(x)
        best = None
        for it in range(3):
            t0 = time.perf_counter()
            enc.encode_host(x, y)
            dt = time.perf_counter() - t0
            if it:
                best = dt if best is None else min(best, dt)
        assert bool((torch.from_numpy(y.view(np.int32)).cuda().view(-1) == want).all())
        out["host_pageable"] = {"ms": round(best * 1e3, 2), "GBps": round(nbytes / best / 1e9, 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
