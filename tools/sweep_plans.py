#!/usr/bin/env python3
"""Time every kernel plan at a given size on one GPU; per-kernel averages from the library's HIP events.
Usage: python tools/sweep_plans.py [--log2k 19] [--block-bytes 4096] [--plans 44,54,...]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import fastecc_amd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--log2k", type=int, default=19)
ap.add_argument("--block-bytes", type=int, default=4096)
ap.add_argument("--plans", default="51,1090,1091,1100,1080,1081")
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--slabs", default="1")
ap.add_argument("--option", action="append", default=[])
ap.add_argument("--pitch", type=int, default=0, help="row pitch in words (0 = contiguous)")
args = ap.parse_args()

k, S = 1 << args.log2k, args.block_bytes // 4
dev = torch.device("cuda", 0)
L = args.pitch or S
data = torch.randint(0, 0xFFF00001, (k * L,), dtype=torch.int64, device=dev).to(torch.int32)
parity = torch.empty_like(data)
enc = fastecc_amd.Encoder(2 * k, k, args.block_bytes)
st = torch.cuda.current_stream().cuda_stream
bytes_per = 2.0 * k * args.block_bytes
for plan, slabs in [(int(p), int(h)) for p in args.plans.split(",") for h in args.slabs.split(",")]:
    enc.set_plan(plan)
    enc.set_option("slabs", slabs)
    if args.pitch:
        enc.set_option("row_pitch_words", args.pitch)
    for kv in args.option:
        enc.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    enc.encode(data, parity, stream=st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        enc.encode(data, parity, stream=st)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    enc.profile(True)
    enc.profile_reset()
    for _ in range(args.steps):
        enc.encode(data, parity, stream=st)
    kern = enc.profile_read()
    enc.profile(False)
    print(json.dumps({"plan": plan, "slabs": slabs, "text": enc.plan(), "ms_per_encode": round(ms, 4), "GBps": round(bytes_per / ms / 1e6, 1),
                      "kernels_avg_ms": {n: round(v[0] / v[1], 4) for n, v in sorted(kern.items())},
                      "launches": {n: v[1] // args.steps for n, v in sorted(kern.items())}}), flush=True)
