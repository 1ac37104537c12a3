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
ap.add_argument("--field", choices=["fff00001", "p61"], default="fff00001")
args = ap.parse_args()
k, S = 1 << args.log2k, args.block_bytes // 4
if args.field == "p61":
    data = torch.randint(0, (1 << 61) - 1, (k * (args.block_bytes // 8),), dtype=torch.int64, device="cuda:0")
else:
    data = torch.randint(0, 0xFFF00001, (k * S,), dtype=torch.int64, device="cuda:0").to(torch.int32)
parity = torch.empty_like(data)
enc = fastecc_amd.Encoder(2 * k, k, args.block_bytes, field=fastecc_amd.FIELD_GF_P61_SQUARED if args.field == "p61" else fastecc_amd.FIELD_GF_FFF00001)
if args.plan:
    enc.set_plan(args.plan)
for _ in range(args.steps):
    enc.encode(data, parity, stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
print(enc.plan())
