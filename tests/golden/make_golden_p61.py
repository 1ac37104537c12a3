#!/usr/bin/env python3
"""Golden vectors for the GF((2^61-1)^2) configuration, from an INDEPENDENT pure-Python statement.

The reference has no code for this field (oracle/fastecc_oracle_p61.h: parity unpinned), so these vectors
do not come from it.  They come from Python big-integer arithmetic applied to the mathematical contract of
the encoder (RS.md / RS.cpp:40-63): the data blocks are the values f(w_N^i) of a polynomial of degree < N,
parity block j is f(w_2N^(2j+1)).  Nothing here shares code with oracle/ or fastecc_amd/.

    python tests/golden/make_golden_p61.py        # rewrites tests/golden/golden_p61.json
"""
import json
import os

P = (1 << 61) - 1


def cmul(x, y):
    return ((x[0] * y[0] - x[1] * y[1]) % P, (x[0] * y[1] + x[1] * y[0]) % P)


def cadd(x, y):
    return ((x[0] + y[0]) % P, (x[1] + y[1]) % P)


def cpow(x, e):
    r = (1, 0)
    while e:
        if e & 1:
            r = cmul(r, x)
        x = cmul(x, x)
        e >>= 1
    return r


def cinv(x):
    return cpow(x, P * P - 2)  # Fermat in GF(p^2)


def generator():
    """Smallest a >= 0 such that a + i is a non-square of GF(p^2) (its 2-power order is then 2^62)."""
    a = 0
    while cpow((a, 1), (P * P - 1) // 2) == (1, 0):
        a += 1
    return (a, 1)


G = generator()
W62 = cpow(G, (P * P - 1) >> 62)


def root(order):
    return cpow(W62, (1 << 62) // order)


def splitmix_words(n, seed):
    s, out = seed, []
    for _ in range(n):
        s = (s + 0x9E3779B97F4A7C15) & (2**64 - 1)
        z = s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & (2**64 - 1)
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & (2**64 - 1)
        z ^= z >> 31
        out.append(z % P)
    return out


def encode(words, N, elems):
    """words: N*elems*2 ints, block-major.  Returns the parity stripe in the same layout."""
    wN, w2N = root(N), root(2 * N)
    inv_wN, inv_N = cinv(wN), cinv((N % P, 0))
    out = [0] * len(words)
    for c in range(elems):
        vals = [(words[(i * elems + c) * 2], words[(i * elems + c) * 2 + 1]) for i in range(N)]
        # interpolate: coef_m = 1/N * sum_i vals_i * w_N^(-i m)
        coef = []
        for m in range(N):
            acc = (0, 0)
            for i in range(N):
                acc = cadd(acc, cmul(vals[i], cpow(inv_wN, i * m)))
            coef.append(cmul(acc, inv_N))
        for j in range(N):
            x = cpow(w2N, 2 * j + 1)
            acc = (0, 0)
            for m in range(N):
                acc = cadd(acc, cmul(coef[m], cpow(x, m)))
            out[(j * elems + c) * 2], out[(j * elems + c) * 2 + 1] = acc
    return out


def coset_generators(N, e):
    """The evaluation cosets of the n = 2^e N codes, in the nesting order of include/fastecc.h: w_2N; w_4N, w_4N^3; w_8N, w_8N^3, w_8N^5, w_8N^7."""
    gens = []
    for j in range(1, e + 1):
        w = root(N << j)
        gens += [cpow(w, c) for c in range(1, 1 << j, 2)]
    return gens


def encode_cosets(words, N, elems, e):
    """n = 2^e N: parity block t N + j = f(g_t w_N^j), f the interpolating polynomial of the data (degree < N), straight from the definition."""
    wN = root(N)
    inv_wN, inv_N = cinv(wN), cinv((N % P, 0))
    gens = coset_generators(N, e)
    out = [0] * (len(gens) * len(words))
    for c in range(elems):
        vals = [(words[(i * elems + c) * 2], words[(i * elems + c) * 2 + 1]) for i in range(N)]
        coef = []
        for m in range(N):
            acc = (0, 0)
            for i in range(N):
                acc = cadd(acc, cmul(vals[i], cpow(inv_wN, i * m)))
            coef.append(cmul(acc, inv_N))
        for t, g in enumerate(gens):
            for j in range(N):
                x = cmul(g, cpow(wN, j))
                acc = (0, 0)
                for m in range(N):
                    acc = cadd(acc, cmul(coef[m], cpow(x, m)))
                out[((t * N + j) * elems + c) * 2], out[((t * N + j) * elems + c) * 2 + 1] = acc
    return out


def main():
    cases = []
    for N, elems, seed in ((2, 1, 0x1234), (4, 2, 0x1234), (16, 1, 7), (32, 2, 99)):
        words = splitmix_words(N * elems * 2, seed)
        cases.append({"N": N, "elems": elems, "seed": seed, "data": [str(w) for w in words],
                      "parity": [str(w) for w in encode(words, N, elems)]})
    coset_cases = []
    # (64, 1, 2, 9): the smallest n = 4k code whose DECODER runs the folded transform (every fourth position: gf61_decode.hip) — the GPU test
    # erases 3k blocks of this codeword and must get this data back
    for N, elems, e, seed in ((2, 1, 2, 5), (4, 2, 2, 6), (8, 1, 3, 7), (16, 1, 2, 8), (64, 1, 2, 9)):
        words = splitmix_words(N * elems * 2, seed)
        coset_cases.append({"N": N, "elems": elems, "e": e, "seed": seed, "data": [str(w) for w in words],
                            "parity": [str(w) for w in encode_cosets(words, N, elems, e)]})
    doc = {
        "p": str(P), "generator": [str(v) for v in G], "w_2^62": [str(v) for v in W62],
        "roots": {str(t): [str(v) for v in root(1 << t)] for t in (1, 2, 3, 4, 8, 16, 19, 20)},
        "inv_2^19": str(pow(1 << 19, P - 2, P)),
        "cases": cases,
        "coset_cases": coset_cases,
    }
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_p61.json")
    with open(path, "w") as f:
        json.dump(doc, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
