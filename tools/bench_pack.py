#!/usr/bin/env python3
"""Time fastecc_pack_blocks / fastecc_unpack_blocks (GF.md:72-104) at the headline stripe: k = 2^19 sectors of
4096 bytes -> 4100-byte blocks.  HIP events around each launch (the library's profile API), HBM-resident data.
One JSON line; roofline = algorithmic bytes (read W words + write W+1, or the reverse) / kernel time vs 8 TB/s."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import fastecc_amd  # noqa: E402


def main():
    k, W, steps = 1 << 19, 1024, 20
    pitch = int(sys.argv[1]) if len(sys.argv) > 1 else W + 1  # e.g. 1056: device rows padded to 4224 bytes
    g = torch.Generator(device="cuda:0").manual_seed(5)
    raw = torch.randint(-(1 << 31), 1 << 31, (k * W,), dtype=torch.int64, device="cuda:0", generator=g).to(torch.int32)
    packed = torch.empty(k * pitch, dtype=torch.int32, device="cuda:0")
    back = torch.empty_like(raw)
    parity = torch.empty_like(packed)
    out = {}
    with fastecc_amd.Encoder(2 * k, k, 4 * (W + 1)) as enc:
        if pitch != W + 1:
            enc.set_option("row_pitch_words", pitch)
        for _ in range(3):
            enc.pack_blocks(raw, packed)
            enc.unpack_blocks(packed, back, count_bad=False)
            enc.encode(packed, parity)
        torch.cuda.synchronize()
        enc.profile(True)
        enc.profile_reset()
        for _ in range(steps):
            enc.pack_blocks(raw, packed)
            enc.unpack_blocks(packed, back, count_bad=False)
            enc.encode(packed, parity)
        prof = enc.profile_read()
        for name, (ms, launches, nbytes) in sorted(prof.items()):
            avg = ms / launches
            out[name] = {"avg_ms": round(avg, 4), "alg_bytes": nbytes // launches,
                         "GBps": round(nbytes / launches / (avg * 1e-3) / 1e9, 1),
                         "frac_of_8TBps": round(nbytes / launches / (avg * 1e-3) / 8e12, 4)}
        assert bool((back == raw).all())
        out["plan"] = enc.plan()
        out["row_pitch_words"] = pitch
    out["workload"] = "k=2^19 sectors x 4096 B -> 4100 B blocks; random 32-bit words (22 % of sectors need recoding)"
    print(json.dumps(out))


if __name__ == "__main__":
    main()
