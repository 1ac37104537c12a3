#!/usr/bin/env python3
"""Minimal driver for profilers: run a few encodes of one plan (no timing, no CPU baseline)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402  (imported first so that the process has a single HIP runtime)

import fastecc_amd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--log2k", type=int, default=19)
ap.add_argument("--block-bytes", type=int, default=4096)
ap.add_argument("--plan", type=int, default=0)
ap.add_argument("--steps", type=int, default=3)
args = ap.parse_args()
k, S = 1 << args.log2k, args.block_bytes // 4
data = torch.randint(0, 0xFFF00001, (k * S,), dtype=torch.int64, device="cuda:0").to(torch.int32)
parity = torch.empty_like(data)
enc = fastecc_amd.Encoder(2 * k, k, args.block_bytes)
if args.plan:
    enc.set_plan(args.plan)
for _ in range(args.steps):
    enc.encode(data, parity, stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
print(enc.plan())
