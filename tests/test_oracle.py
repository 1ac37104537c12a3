"""CPU tests: the oracle restatement against golden vectors, the published KAT and (when prebuilt) the
unmodified reference.  These pin the checker that the -m gpu parity tests rely on."""
import os

import numpy as np
import pytest

P = 0xFFF00001


def test_field_ops_against_bigint(oracle):
    rng = np.random.default_rng(1)
    xs = [0, 1, 2, P - 1, P - 2, 0x000FFFFF, 0x00100000, 0xFFEFFFFF] + rng.integers(0, P, 200).tolist()
    ys = [0, 1, P - 1, 19, 0xFDB9DED3] + rng.integers(0, P, 40).tolist()
    for x in xs:
        for y in ys:
            assert oracle.gf_add(x, y) == (x + y) % P
            assert oracle.gf_sub(x, y) == (x - y) % P
            assert oracle.gf_mul(x, y) == (x * y) % P
            assert oracle.gf_mul_wide(x, y) == (x * y) % P


def test_constants_survey_appendix_c(oracle):
    # SURVEY.md Appendix C (computed from GF_Root / GF_Inv of the reference)
    assert oracle.gf_root(2) == P - 1
    assert oracle.gf_root(4) == 4256816851
    assert oracle.gf_root(1 << 7) == 955468005
    assert oracle.gf_root(1 << 19) == 1390037254
    assert oracle.gf_root(1 << 20) == 3156611342
    assert oracle.gf_inv(1 << 19) == 4293910531
    assert oracle.gf_pow(19, P - 1) == 1
    assert oracle.gf_mul(oracle.gf_root(1 << 20), oracle.gf_root(1 << 20)) == oracle.gf_root(1 << 19)


def test_field_ops_against_reference(oracle, reference):
    rng = np.random.default_rng(2)
    for x, y in rng.integers(0, P, (500, 2)).tolist() + [[0, 0], [P - 1, P - 1], [P - 1, 1], [0, P - 1]]:
        assert oracle.gf_add(x, y) == reference.gf_add(x, y)
        assert oracle.gf_sub(x, y) == reference.gf_sub(x, y)
        assert oracle.gf_mul(x, y) == reference.gf_mul(x, y)
    for order in [2, 4, 256, 1 << 15, 1 << 20]:
        assert oracle.gf_root(order) == reference.gf_root(order)


def test_published_kat_forward_ntt(oracle, golden_hashes):
    # Benchmarks.md:491,499,507: `ntt {q,o,n} 20 32`
    kat = golden_hashes["published_kat"]["ntt_fwd_2^20x32B_linear"]
    x = oracle.fill_linear(1 << 20, 8)
    assert oracle.hash(x) == kat["hash_input"]
    assert oracle.hash(oracle.ntt_fast(x)) == kat["hash_output"]


def test_three_ntt_formulations_agree(oracle):
    rng = np.random.default_rng(3)
    for N, S in [(2, 3), (8, 5), (64, 4), (256, 2)]:
        x = rng.integers(0, P, (N, S)).astype(np.uint32)
        for inv in (False, True):
            a = oracle.slow_ntt(x, inv)
            assert np.array_equal(a, oracle.ntt(x, inv))
            assert np.array_equal(a, oracle.ntt_fast(x, inv))
        # inverse(forward(x)) / N == x   (main.cpp:286-299)
        back = oracle.scale_blocks(oracle.ntt(oracle.ntt(x), True), oracle.gf_inv(N), 1)
        assert np.array_equal(back, x)


def test_encode_matches_mathematical_contract(oracle):
    # parity[j] = f(w_2N^(2j+1)), f interpolating the data at powers of w_N
    rng = np.random.default_rng(4)
    for N, S in [(2, 2), (4, 4), (16, 3), (32, 1)]:
        x = rng.integers(0, P, (N, S)).astype(np.uint32)
        assert np.array_equal(oracle.encode(x), oracle.encode_by_definition(x))
        assert np.array_equal(oracle.encode_fast(x), oracle.encode_by_definition(x))


def test_golden_hash_cases(oracle, golden_hashes):
    for c in golden_hashes["cases"]:
        if c["log2N"] > 15:
            continue
        N, S = 1 << c["log2N"], c["block_bytes"] // 4
        x = oracle.fill_linear(N, S) if c["input"] == "linear" else oracle.fill_splitmix(N, S, golden_hashes["splitmix_seed"])
        assert oracle.hash(x) == c["hash_input"]
        par = oracle.encode_fast(x)
        assert oracle.hash(par) == c["hash_parity"], c
        assert par[0, :4].tolist() == c["parity_0_0_4"]
        assert int(par[1, 0]) == c["parity_1_0"] and int(par[-1, -1]) == c["parity_last_last"]
        if "hash_ntt_fwd" in c:
            assert oracle.hash(oracle.ntt_fast(x, False)) == c["hash_ntt_fwd"]
            assert oracle.hash(oracle.ntt_fast(x, True)) == c["hash_ntt_inv"]


def test_golden_vectors(oracle, golden_vectors):
    keys = sorted(k[:-3] for k in golden_vectors if k.endswith("_in"))
    assert keys
    for k in keys:
        x = golden_vectors[k + "_in"]
        assert np.array_equal(oracle.encode(x), golden_vectors[k + "_parity"]), k
        assert np.array_equal(oracle.ntt(x, False), golden_vectors[k + "_fwd"]), k
        assert np.array_equal(oracle.ntt(x, True), golden_vectors[k + "_inv"]), k


def test_oracle_against_reference_encode(oracle, reference):
    rng = np.random.default_rng(5)
    for N, S in [(2, 1), (4, 7), (128, 1024), (1024, 513), (4096, 33)]:
        x = rng.integers(0, P, (N, S)).astype(np.uint32)
        assert np.array_equal(oracle.encode_fast(x), reference.encode(x)), (N, S)
        assert reference.hash(x) == oracle.hash(x)


def test_linearity_and_edge_inputs(oracle):
    rng = np.random.default_rng(6)
    N, S = 64, 6
    a = rng.integers(0, P, (N, S)).astype(np.uint32)
    b = rng.integers(0, P, (N, S)).astype(np.uint32)
    ab = ((a.astype(np.uint64) + b) % P).astype(np.uint32)
    ea, eb, eab = oracle.encode(a), oracle.encode(b), oracle.encode(ab)
    assert np.array_equal(((ea.astype(np.uint64) + eb) % P).astype(np.uint32), eab)
    z = np.zeros((N, S), dtype=np.uint32)
    assert not oracle.encode(z).any()
    # a constant stripe is the constant polynomial: parity == data
    c = np.full((N, S), P - 1, dtype=np.uint32)
    assert np.array_equal(oracle.encode(c), c)


# ------------------------------------------------------------------------------------------------
# GF((2^61-1)^2) oracle (parity unpinned upstream): pinned to independent big-integer vectors
# ------------------------------------------------------------------------------------------------
def test_p61_oracle_against_independent_golden():
    import json
    import oracle as orc
    o = orc.OracleP61()
    doc = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_p61.json")))
    assert o.root(1 << 62) == tuple(int(v) for v in doc["w_2^62"])
    for t, v in doc["roots"].items():
        assert o.root(1 << int(t)) == tuple(int(x) for x in v)
    assert o.root(4) == (0, 1) and o.root(8) == (1 << 30, 1 << 30)  # w_4 = i, w_8 = 2^30 (1 + i)
    assert o.root(3) == (0, 0) and o.root(1 << 63) == (0, 0)
    for case in doc["cases"]:
        N = case["N"]
        x = np.array([int(w) for w in case["data"]], dtype=np.uint64).reshape(N, -1)
        want = np.array([int(w) for w in case["parity"]], dtype=np.uint64).reshape(N, -1)
        assert (o.encode(x) == want).all()
        assert (o.encode_by_definition(x) == want).all()


def test_p61_oracle_multi_coset_composition_against_independent_golden():
    """n = 4k / 8k over the 64-bit field: the oracle's composition (iNTT, block i *= g^i / N, NTT per coset generator g, the nesting order of
    include/fastecc.h) gives the big-integer vectors computed straight from the definition f(g w_k^j) (tests/golden/make_golden_p61.py)."""
    import json
    import oracle as orc
    o = orc.OracleP61()
    doc = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_p61.json")))
    assert len(doc["coset_cases"]) >= 4
    for case in doc["coset_cases"]:
        N, e = case["N"], case["e"]
        x = np.array([int(w) for w in case["data"]], dtype=np.uint64).reshape(N, -1)
        want = np.array([int(w) for w in case["parity"]], dtype=np.uint64).reshape(((1 << e) - 1) * N, -1)
        gens = []
        for j in range(1, e + 1):
            w = o.root(N << j)
            gens += [o.cpow(w, c) for c in range(1, 1 << j, 2)]
        coef = o.ntt(x, inverse=True)
        inv_n = o.cinv((N % ((1 << 61) - 1), 0))
        got = np.concatenate([o.ntt(o.scale_blocks(coef, inv_n, g)) for g in gens])
        assert (got == want).all(), (N, e)
        assert (got[:N] == o.encode(x)).all()  # the codes nest: coset 0 is the (2k,k) parity


def test_p61_oracle_fast_transform_is_the_definition():
    import oracle as orc
    o = orc.OracleP61()
    x = o.fill_splitmix(64, 3, 42)
    for inverse in (False, True):
        assert (o.ntt(x, inverse) == o.slow_ntt(x, inverse)).all()
    # codeword property: data (even points) and parity (odd points) interleaved are the values of one polynomial
    # of degree < N on the 2N-th roots of unity, so the inverse 2N-transform has a zero upper half
    N = 64
    par = o.encode(x)
    word = np.empty((2 * N, x.shape[1]), dtype=np.uint64)
    word[0::2], word[1::2] = x, par
    coef = o.ntt(word, inverse=True)
    assert (coef[N:] == 0).all() and (coef[:N] != 0).any()


# ------------------------------------------------------------------------------------------------
# data packing (GF.md:72-104): the C restatement against the independent pure-Python vectors
# ------------------------------------------------------------------------------------------------
def test_pack_oracle_against_independent_golden(oracle):
    import json
    doc = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_pack.json")))
    assert len(doc["cases"]) >= 10
    for case in doc["cases"]:
        raw = np.array(case["raw"], dtype=np.uint32)[None, :]
        want = np.array(case["packed"], dtype=np.uint32)[None, :]
        got = oracle.pack_blocks(raw)
        assert (got == want).all(), case["name"]
        assert (got < P).all()
        back, bad = oracle.unpack_blocks(got)
        assert bad == 0 and (back == raw).all()


def test_pack_oracle_properties(oracle):
    rng = np.random.default_rng(1)
    for W in (1, 7, 64, 1000, 1024):
        raw = rng.integers(0, 1 << 32, size=(50, W), dtype=np.uint64).astype(np.uint32)
        raw[::3] |= np.uint32(0xFFF00000) * (rng.random((len(raw[::3]), W)) < 0.1).astype(np.uint32)
        packed = oracle.pack_blocks(raw)
        assert (packed < P).all() and set(np.unique(packed[:, W])) <= {0, 1}
        assert ((packed[:, :W] & 0xFFFFF) == (raw & 0xFFFFF)).all()      # low 20 bits never move
        back, bad = oracle.unpack_blocks(packed)
        assert bad == 0 and (back == raw).all()
    junk = np.full((1, 9), 0, dtype=np.uint32)
    junk[0, 8] = 7
    assert oracle.unpack_blocks(junk)[1] == 1


def test_coset_composition_is_polynomial_evaluation(oracle):
    """iNTT, block i *= g^i / N, NTT  ==  f(g * w_N^j) for any g (here the generators of the n = 4k, 8k parity cosets)."""
    N, S = 8, 2
    x = oracle.fill_splitmix(N, S, 77)
    wN, inv_n = oracle.gf_root(N), oracle.gf_inv(N)
    coef = oracle.scale_blocks(oracle.slow_ntt(x, inverse=True), inv_n, 1)  # f's coefficients
    for order, c in ((2 * N, 1), (4 * N, 1), (4 * N, 3), (8 * N, 5)):
        g = oracle.gf_pow(oracle.gf_root(order), c)
        got = oracle.ntt_fast(oracle.scale_blocks(oracle.ntt_fast(x, inverse=True), inv_n, g))
        for j in range(N):
            pt = oracle.gf_mul(g, oracle.gf_pow(wN, j))
            for s in range(S):
                acc = 0
                for i in range(N):
                    acc = (acc + int(coef[i, s]) * oracle.gf_pow(pt, i)) % P
                assert got[j, s] == acc


def test_lagrange_decoder_recovers_what_the_encoder_wrote(oracle):
    """orc_decode (checker of fastecc_decode) against the pinned encoder: any N survivors give the data back."""
    rng = np.random.default_rng(8)
    for N, S in ((2, 3), (8, 5), (32, 4)):
        x = oracle.fill_splitmix(N, S, N)
        par = oracle.encode(x)
        for trial in range(4):
            lost = rng.permutation(2 * N)[:N]
            dp, pp = np.ones(N, np.uint8), np.ones(N, np.uint8)
            dp[lost[lost < N]] = 0
            pp[lost[lost >= N] - N] = 0
            damaged = x.copy()
            damaged[dp == 0] = 0xFFFFFFFF
            assert (oracle.decode(damaged, par, dp, pp) == x).all()
        dp = np.zeros(N, np.uint8)
        pp = np.ones(N, np.uint8)
        pp[0] = 0
        assert oracle.decode(x, par, dp, pp) is None  # N - 1 survivors


# ------------------------------------------------------------------------------------------------
# mixed-radix orders (NTT.md:43-46): the oracle's transforms of order q * 2^m against the reference
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N", [3, 6, 12, 5, 10, 40, 7, 28, 9, 18, 72, 96, 160, 13, 26, 104, 15, 30, 120, 42, 70, 156, 90, 252, 130, 182, 210, 468])
def test_mixed_radix_transform_matches_the_definition(oracle, N):
    """orc_ntt_mixed (odd factor outermost + radix-2) == the O(N^2) definition orc_slow_ntt, both directions."""
    x = np.random.default_rng(N).integers(0, P, size=(N, 3), dtype=np.uint64).astype(np.uint32)
    for inverse in (False, True):
        assert np.array_equal(oracle.ntt_mixed(x, inverse), oracle.slow_ntt(x, inverse)), (N, inverse)
    assert np.array_equal(oracle.encode_mixed(x), oracle.encode_slow(x))


@pytest.mark.parametrize("N", [3, 12, 24, 9, 36, 5, 20, 7, 14, 13, 52, 15, 60, 42, 70, 78, 90, 126, 130, 182, 210, 234])
def test_mixed_radix_is_pinned_to_the_reference(oracle, reference, N):
    """Slow_NTT (ntt.cpp:451-483) is the one reference transform that accepts a non-power-of-two order; the encode is the
    RS.cpp:40-63 composition around it.  Both must agree with the oracle's fast mixed-radix path."""
    from oracle import Reference
    x = np.random.default_rng(100 + N).integers(0, P, size=(N, 2), dtype=np.uint64).astype(np.uint32)
    for inverse in (False, True):
        assert np.array_equal(reference.ntt(x, inverse, Reference.SLOW), oracle.ntt_mixed(x, inverse)), (N, inverse)
    # RS.cpp:41-63 with Slow_NTT: inverse transform, block i *= root(2N)^i / N, forward transform
    c = reference.ntt(x, True, Reference.SLOW).astype(object)
    w2n, inv_n = reference.gf_root(2 * N), reference.gf_inv(N)
    for i in range(N):
        c[i] = (c[i] * (inv_n * pow(w2n, i, P) % P)) % P
    want = reference.ntt(c.astype(np.uint32), False, Reference.SLOW)
    assert np.array_equal(oracle.encode_mixed(x), want)


def test_reference_codelets_of_order_3_and_9(oracle, reference):
    """NTT3 / NTT9 (ntt.cpp:25-44, 113-146) compute the order-3 / order-9 transform of the definition: the values the
    odd-radix pass of the HIP path is built on.  (NTT9 leaves its outputs in natural order: its last step transposes.)"""
    if not hasattr(reference.lib, "ref_small_ntt"):
        pytest.skip("oracle/_ref predates ref_small_ntt: rebuild it where /root/reference exists")
    rng = np.random.default_rng(39)
    for order in (3, 9):
        for inverse in (False, True):
            f = rng.integers(0, P, size=order, dtype=np.uint64).astype(np.uint32)
            want = oracle.slow_ntt(f.reshape(order, 1), inverse).reshape(-1)
            assert np.array_equal(reference.small_ntt(f, inverse), want), (order, inverse)


def test_reference_codelets_of_order_2_and_4(oracle, reference):
    """NTT2 / NTT4 (ntt.cpp:16-22, 50-62) — the codelets every power-of-two transform of the reference bottoms out in — compute the
    order-2 / order-4 transform of the definition in natural order (NTT4: after its closing swap), and so do the oracle's own
    transforms at N = 2, 4 (pinned to the reference's code rather than to the mathematics)."""
    if not hasattr(reference.lib, "ref_small_ntt") or reference.lib.ref_small_ntt(np.zeros(4, np.uint32), 4, 0) != 0:
        pytest.skip("oracle/_ref predates the order-2 / order-4 entries of ref_small_ntt: rebuild it where /root/reference exists")
    from oracle import Reference
    rng = np.random.default_rng(24)
    edge = np.array([0, 1, P - 1, P - 2], dtype=np.uint32)
    for order in (2, 4):
        for inverse in (False, True):
            for f in [rng.integers(0, P, size=order, dtype=np.uint64).astype(np.uint32) for _ in range(20)] + [edge[:order], edge[::-1][:order].copy()]:
                got = reference.small_ntt(f, inverse)
                assert np.array_equal(got, oracle.slow_ntt(f.reshape(order, 1), inverse).reshape(-1)), (order, inverse)
                assert np.array_equal(got, oracle.ntt_fast(f.reshape(order, 1), inverse).reshape(-1)), (order, inverse)
                assert np.array_equal(got, reference.ntt(f.reshape(order, 1), inverse, Reference.MFA).reshape(-1)), (order, inverse)


def test_p61_oracle_decoder_round_trip():
    """orc61_decode (O(N^2) Lagrange) restores what orc61_encode produced: the checker of the GF(p61^2) HIP decoder."""
    from oracle import OracleP61
    o = OracleP61()
    rng = np.random.default_rng(61)
    for N, elems in ((2, 1), (8, 3), (32, 2)):
        x = o.fill_splitmix(N, elems, 0x61 + N)
        par = o.encode(x)
        lost = rng.permutation(2 * N)[:N]
        dp, pp = np.ones(N, np.uint8), np.ones(N, np.uint8)
        dp[lost[lost < N]] = 0
        pp[lost[lost >= N] - N] = 0
        damaged = x.copy()
        damaged[dp == 0] = 12345
        assert np.array_equal(o.decode(damaged, par, dp, pp), x)


def test_bench_generates_the_splitmix_stripe_window_by_window(oracle):
    """bench.py's N > 1 line checks the exchanged parity against the reference's hash of the splitmix64(0x1234) stripe; every rank generates its
    own window of that stripe on its device with 64-bit integer tensor arithmetic (bench.splitmix_window).  It must be the oracle's stripe."""
    import importlib.util
    import os
    import sys
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(bench)
    finally:
        sys.argv = argv
    N, S = 96, 40
    want = oracle.fill_splitmix(N, S, 0x1234)
    cpu = torch.device("cpu")
    assert np.array_equal(bench.splitmix_window(cpu, S, 0, N, 0, S).numpy().view(np.uint32), want)
    assert np.array_equal(bench.splitmix_window(cpu, S, 17, 30, 8, 16).numpy().view(np.uint32), want[17:47, 8:24])
    assert np.array_equal(bench.splitmix_window(cpu, S, 0, N, 39, 1, seed=0x1234).numpy().view(np.uint32), want[:, 39:40])
