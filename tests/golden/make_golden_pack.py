#!/usr/bin/env python3
"""Golden vectors for the data packing of GF.md:72-104, from an INDEPENDENT pure-Python statement.

Upstream gives prose only, so these vectors pin OUR format (include/fastecc.h), not the reference.  The
unpacker below is written from the prose ("Decoding algorithm ...", GF.md:80-85) and the packer is the
obvious inverse; neither shares code with oracle/ or fastecc_amd/.

    python tests/golden/make_golden_pack.py        # rewrites tests/golden/golden_pack.json
"""
import json
import os
import random

P = 0xFFF00001


def pack(raw):
    digits = [w >> 20 for w in raw]
    pos = [j for j, d in enumerate(digits) if d == 0xFFF]
    if not pos:
        return list(raw) + [0]
    new = [p | (0x400 if t + 1 < len(pos) else 0) for t, p in enumerate(pos)] + [d for d in digits if d != 0xFFF]
    return [(d << 20) | (w & 0xFFFFF) for d, w in zip(new, raw)] + [1]


def unpack(packed):
    *body, flag = packed
    if flag == 0:
        return list(body)
    digits = [w >> 20 for w in body]
    pos, t = [], 0
    while True:  # "first entry holds index of the first 0xFFF (10 bits), plus the flag (1 - there are more)"
        pos.append(digits[t] & 0x3FF)
        more = digits[t] & 0x400
        t += 1
        if not more:
            break
    rest = iter(digits[t:])  # "after the flag 0, remaining input items contain values of remaining output elements"
    out = [0xFFF if j in set(pos) else next(rest) for j in range(len(body))]
    return [(d << 20) | (w & 0xFFFFF) for d, w in zip(out, body)]


def main():
    rnd = random.Random(20240925)
    cases = []

    def add(name, raw):
        packed = pack(raw)
        assert unpack(packed) == raw and all(w < P for w in packed)
        cases.append({"name": name, "raw": raw, "packed": packed})

    add("no_fff_8", [rnd.randrange(0xFFF00000) for _ in range(8)])
    add("one_fff_8", [5, 0xFFF12345, 7, 9, 11, 13, 15, 17])
    add("all_fff_4", [0xFFFFFFFF, 0xFFF00000, 0xFFF00001, 0xFFFABCDE])
    add("first_and_last_16", [0xFFFFFFFF] + [rnd.randrange(1 << 32) & 0xFFEFFFFF for _ in range(14)] + [0xFFF00001])
    add("single_word", [0xFFF55555])
    add("single_word_plain", [0x12345678])
    w = [rnd.randrange(1 << 32) for _ in range(1024)]
    for j in (0, 1, 63, 64, 65, 511, 512, 1000, 1023):
        w[j] |= 0xFFF00000
    add("full_1024_sparse", w)
    add("full_1024_dense", [(0xFFF00000 | rnd.randrange(1 << 20)) if rnd.random() < 0.6 else rnd.randrange(1 << 32) for _ in range(1024)])
    add("full_1024_all", [0xFFF00000 | rnd.randrange(1 << 20) for _ in range(1024)])
    add("ragged_1000", [(0xFFF00000 | j) if j % 97 == 0 else rnd.randrange(1 << 32) for j in range(1000)])
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_pack.json")
    with open(path, "w") as f:
        json.dump({"P": P, "cases": cases}, f)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
