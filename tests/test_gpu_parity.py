"""GPU parity tests (-m gpu): every call goes through the C ABI of libfastecc_hip.so and is compared
bit-for-bit with the CPU oracle, the committed golden fixtures, or a size-independent property.

Bar: bit-exact (integer arithmetic mod p = 0xFFF00001); no tolerance anywhere."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

P = 0xFFF00001


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    return torch


@pytest.fixture(scope="module")
def fe(hip_lib):
    import fastecc_amd
    return fastecc_amd


def to_dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to("cuda:0")


def to_host(t):
    return t.cpu().numpy().view(np.uint32)


def rand_stripe(rng, N, S):
    x = rng.integers(0, P, size=(N, S), dtype=np.uint64).astype(np.uint32)
    # edge values: 0, 1, p-1 and the wrap-around neighbourhood of 2^32 mod p
    flat = x.reshape(-1)
    edge = [0, 1, P - 1, P - 2, 0x000FFFFF, 0x00100000, 0xFFEFFFFF]
    flat[: min(len(edge), flat.size)] = edge[: flat.size]
    return x


# ------------------------------------------------------------------------------------------------
# field kernels (GF(p).cpp:37-48, 110-127)
# ------------------------------------------------------------------------------------------------
def test_gf_kernels_match_oracle(torch_cuda, fe, oracle):
    torch = torch_cuda
    rng = np.random.default_rng(11)
    edge = np.array([0, 1, 2, P - 1, P - 2, 0x000FFFFF, 0x00100000, 0x00100001, 0xFFEFFFFF, 0xFFF00000], dtype=np.uint32)
    x = np.concatenate([np.repeat(edge, edge.size), rng.integers(0, P, 200_000).astype(np.uint32)])
    y = np.concatenate([np.tile(edge, edge.size), rng.integers(0, P, 200_000).astype(np.uint32)])
    dx, dy = to_dev(torch, x), to_dev(torch, y)
    out = torch.empty_like(dx)
    X, Y = x.astype(object), y.astype(object)
    want = {"add": (X + Y) % P, "sub": (X - Y) % P, "mul": (X * Y) % P, "mul_mont": (X * Y) % P}
    with fe.Encoder(4, 2, 4) as enc:
        for op in ("add", "sub", "mul", "mul_mont"):
            enc.gf_binary(op, dx, dy, out, x.size)
            torch.cuda.synchronize()
            got = to_host(out)
            assert np.array_equal(got.astype(object), want[op]), op
    # and the same through the oracle's C routines on a sample (pins oracle == bigint == GPU)
    for i in range(0, 300):
        assert oracle.gf_mul(int(x[i]), int(y[i])) == int(want["mul"][i])


# ------------------------------------------------------------------------------------------------
# per-block twiddle multiply (RS.cpp:51-59, ntt.cpp:421-431)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,S", [(2, 1), (8, 3), (64, 513), (256, 1024)])
def test_scale_blocks(torch_cuda, fe, oracle, N, S):
    torch = torch_cuda
    x = rand_stripe(np.random.default_rng(N * 7 + S), N, S)
    scale, base = oracle.gf_inv(N), oracle.gf_root(2 * N)
    d = to_dev(torch, x)
    with fe.Encoder(2 * N, N, 4 * S) as enc:
        enc.scale_blocks(d, scale, base)
        torch.cuda.synchronize()
    assert np.array_equal(to_host(d), oracle.scale_blocks(x, scale, base))


# ------------------------------------------------------------------------------------------------
# stand-alone transform (MFA_NTT / Rec_NTT semantics, natural order in and out)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("log2n,S", [(1, 1), (2, 4), (3, 3), (5, 513), (6, 64), (9, 7), (10, 1024), (13, 6)])
def test_ntt_matches_oracle(torch_cuda, fe, oracle, log2n, S):
    torch = torch_cuda
    N = 1 << log2n
    x = rand_stripe(np.random.default_rng(100 + log2n), N, S)
    with fe.Encoder(2 * N, N, 4 * S) as enc:
        for inverse in (False, True):
            d = to_dev(torch, x)
            enc.ntt(d, inverse)
            torch.cuda.synchronize()
            assert np.array_equal(to_host(d), oracle.ntt_fast(x, inverse)), (log2n, S, inverse)
        # round trip: inverse(forward(x)) * 1/N == x  (main.cpp:286-299)
        d = to_dev(torch, x)
        enc.ntt(d, False)
        enc.ntt(d, True)
        enc.scale_blocks(d, oracle.gf_inv(N), 1)
        torch.cuda.synchronize()
        assert np.array_equal(to_host(d), x)


def test_ntt_golden_vectors(torch_cuda, fe, golden_vectors):
    torch = torch_cuda
    for key in sorted(k[:-3] for k in golden_vectors if k.endswith("_in")):
        x = golden_vectors[key + "_in"]
        N, S = x.shape
        with fe.Encoder(2 * N, N, 4 * S) as enc:
            for inverse, suffix in ((False, "_fwd"), (True, "_inv")):
                d = to_dev(torch, x)
                enc.ntt(d, inverse)
                torch.cuda.synchronize()
                assert np.array_equal(to_host(d), golden_vectors[key + suffix]), (key, suffix)


# ------------------------------------------------------------------------------------------------
# encode (RS.cpp:39-67)
# ------------------------------------------------------------------------------------------------
def test_encode_golden_vectors(torch_cuda, fe, golden_vectors):
    torch = torch_cuda
    for key in sorted(k[:-3] for k in golden_vectors if k.endswith("_in")):
        x = golden_vectors[key + "_in"]
        N, S = x.shape
        d = to_dev(torch, x)
        out = torch.empty_like(d)
        with fe.Encoder(2 * N, N, 4 * S) as enc:
            enc.encode(d, out)
            torch.cuda.synchronize()
        assert np.array_equal(to_host(out), golden_vectors[key + "_parity"]), key
        assert np.array_equal(to_host(d), x), "out-of-place encode must not touch its input"


@pytest.mark.parametrize("log2n", list(range(1, 15)))
@pytest.mark.parametrize("S", [1, 6, 513, 1024])
def test_encode_matches_oracle(torch_cuda, fe, oracle, log2n, S):
    torch = torch_cuda
    N = 1 << log2n
    if N * S > (1 << 22):
        pytest.skip("kept for the golden-hash test")
    x = rand_stripe(np.random.default_rng(1000 * log2n + S), N, S)
    d = to_dev(torch, x)
    with fe.Encoder(2 * N, N, 4 * S) as enc:
        enc.encode(d)  # in place, like the reference
        torch.cuda.synchronize()
    assert np.array_equal(to_host(d), oracle.encode_fast(x))


@pytest.mark.parametrize("plan", [11, 14, 21, 22, 24, 31, 32, 34, 41, 42, 44, 51, 52, 54,
                                  1060, 1061, 1070, 1071, 1080, 1081, 1090, 1091, 1094, 1100, 1104, 1105, 2080, 2090, 2100, 3080, 3090, 3100, 4080, 4090, 4100])
def test_every_plan_is_bit_exact(torch_cuda, fe, oracle, plan):
    """Register plans (levels-per-pass * 10 + words-per-lane) and LDS-tiled plans (1000 + 10*mid_levels
    + wide flag, fastecc_set_plan): all must give identical parity."""
    torch = torch_cuda
    for log2n, S in [(4, 4), (6, 3), (7, 70), (9, 12), (10, 33), (11, 64), (13, 40), (15, 16), (17, 5)]:
        N = 1 << log2n
        x = rand_stripe(np.random.default_rng(plan * 31 + log2n), N, S)
        want = oracle.encode_fast(x)
        d = to_dev(torch, x)
        with fe.Encoder(2 * N, N, 4 * S) as enc:
            enc.set_plan(plan)
            enc.encode(d)
            torch.cuda.synchronize()
        assert np.array_equal(to_host(d), want), (plan, log2n, enc.plan())


def test_encode_golden_hashes(torch_cuda, fe, oracle, golden_hashes):
    """BASELINE configs 1 and 2 ((256,128) and (2^16,2^15), 4 KB blocks) + ragged sizes, against values
    recorded from the unmodified reference (tests/golden/make_golden.py)."""
    torch = torch_cuda
    for c in golden_hashes["cases"]:
        N, S = 1 << c["log2N"], c["block_bytes"] // 4
        x = oracle.fill_linear(N, S) if c["input"] == "linear" else oracle.fill_splitmix(N, S, golden_hashes["splitmix_seed"])
        assert oracle.hash(x) == c["hash_input"]
        d = to_dev(torch, x)
        with fe.Encoder(2 * N, N, 4 * S) as enc:
            enc.encode(d)
            torch.cuda.synchronize()
        par = to_host(d)
        assert oracle.hash(par) == c["hash_parity"], c
        assert par[0, :4].tolist() == c["parity_0_0_4"]
        assert int(par[1, 0]) == c["parity_1_0"] and int(par[-1, -1]) == c["parity_last_last"]


def test_host_memory_and_block_pointer_forms(torch_cuda, fe, oracle):
    N, S = 256, 513
    x = rand_stripe(np.random.default_rng(77), N, S)
    want = oracle.encode_fast(x)
    with fe.Encoder(2 * N, N, 4 * S) as enc:
        # FASTECC_MEM_HOST: plain host buffers in, parity out
        out = np.empty_like(x)
        enc.encode_host(x, out)
        assert np.array_equal(out, want)
        # the reference's T** form: k separately allocated blocks, in place (RS.cpp:31-33)
        blocks = [np.ascontiguousarray(x[i]).copy() for i in range(N)]
        enc.encode_blocks([b.ctypes.data for b in blocks])
        assert np.array_equal(np.stack(blocks), want)
        # host-memory transform
        y = x.copy()
        enc.ntt(y, inverse=True, mem=fe.MEM_HOST)
        assert np.array_equal(y, oracle.ntt_fast(x, True))


def test_host_memory_results_through_the_staging_ring(torch_cuda, fe, oracle):
    """Pageable results of 64 MiB and more leave HBM through the pinned ring that helper threads empty (host_stage.hip stage_download): a
    ragged size (no multiple of a slot, nor of the helpers' pieces), encode and transform, twice on one context (slots reused)."""
    N, S = 1 << 12, 4099  # 67.2 MB per stripe
    x = rand_stripe(np.random.default_rng(4242), N, S)
    want = oracle.encode_fast(x)
    with fe.Encoder(2 * N, N, 4 * S) as enc:
        for _ in range(2):
            out = np.full_like(x, 0xA5A5A5A5)
            enc.encode_host(x, out)
            assert np.array_equal(out, want)
        y = x.copy()
        enc.ntt(y, inverse=False, mem=fe.MEM_HOST)
        assert np.array_equal(y, oracle.ntt_fast(x, False))


def test_host_memory_encode_pipelined_through_both_rings(torch_cuda, fe, oracle):
    """A 256 MiB stripe in pageable memory: column slab h goes up through one ring of pinned slots while slab h - 1 comes down through the
    other (host_stage.hip encode_host_pageable).  Same parity as the oracle, as the one-after-the-other sequence (option host_pipeline = 0, the default) and as
    other slab counts; the data is left alone."""
    N, S = 1 << 16, 1024
    x = rand_stripe(np.random.default_rng(99), N, S)
    keep = x.copy()
    want = oracle.encode_fast(x)
    with fe.Encoder(2 * N, N, 4 * S) as enc:
        enc.set_option("host_pipeline", 1)
        for slabs in (8, 2, 4):
            enc.set_option("host_slabs", slabs)
            out = np.full_like(x, 0x5A5A5A5A)
            enc.encode_host(x, out)
            assert np.array_equal(out, want), slabs
        enc.set_option("host_pipeline", 0)
        out = np.full_like(x, 0x5A5A5A5A)
        enc.encode_host(x, out)
        assert np.array_equal(out, want)
    assert np.array_equal(x, keep)


def test_error_codes_on_device(torch_cuda, fe):
    torch = torch_cuda
    with fe.Encoder(16, 8, 16) as enc:
        d = torch.zeros(8 * 4, dtype=torch.int32, device="cuda:0")
        with pytest.raises(fe.FastEccError) as ei:
            enc.encode(0, d)
        assert ei.value.code == fe.E_INVAL
        with pytest.raises(fe.FastEccError):
            enc.encode(d.data_ptr() + 2, d)  # misaligned
        with pytest.raises(fe.FastEccError):
            enc.scale_blocks(d, P, 1)  # scale not a field element
        with pytest.raises(fe.FastEccError):
            enc.set_plan(99)
    with pytest.raises(fe.FastEccError) as ei:
        fe.Encoder(16, 8, 16, device=4096)
    assert ei.value.code == fe.E_INVAL


def test_streams_are_respected(torch_cuda, fe, oracle):
    torch = torch_cuda
    N, S = 1024, 256
    x = rand_stripe(np.random.default_rng(5), N, S)
    want = oracle.encode_fast(x)
    st = torch.cuda.Stream()
    with fe.Encoder(2 * N, N, 4 * S) as enc:
        with torch.cuda.stream(st):
            d = to_dev(torch, x)
            out = torch.empty_like(d)
            enc.encode(d, out, stream=st.cuda_stream)
        st.synchronize()
        assert np.array_equal(to_host(out), want)


# ------------------------------------------------------------------------------------------------
# headline size (2^20, 2^19), 4 KB blocks: golden hash + size-independent properties
# ------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def headline(torch_cuda, fe):
    N, S = 1 << 19, 1024
    enc = fe.Encoder(2 * N, N, 4 * S)
    yield enc, N, S
    enc.close()


def test_headline_golden_hash_splitmix(torch_cuda, headline, oracle, golden_hashes):
    torch = torch_cuda
    enc, N, S = headline
    c = [g for g in golden_hashes["survey_appendix_b"] if g["input"] == "splitmix"][0]
    x = oracle.fill_splitmix(N, S, golden_hashes["splitmix_seed"])
    assert oracle.hash(x) == c["hash_input"]
    d = to_dev(torch, x)
    del x
    enc.encode(d)
    torch.cuda.synchronize()
    par = to_host(d)
    assert par[0, :4].tolist() == c["parity_0_0_4"]
    assert int(par[1, 0]) == c["parity_1_0"] and int(par[-1, -1]) == c["parity_last_last"]
    assert oracle.hash(par) == c["hash_parity"]


def test_headline_golden_hash_linear(torch_cuda, headline, oracle, golden_hashes):
    torch = torch_cuda
    enc, N, S = headline
    c = [g for g in golden_hashes["survey_appendix_b"] if g["input"] == "linear"][0]
    # i % p on the device (RS.cpp:28-29); 2^29 < p so it is just iota
    d = torch.arange(N * S, dtype=torch.int32, device="cuda:0")
    enc.encode(d)
    torch.cuda.synchronize()
    par = to_host(d).reshape(N, S)
    assert par[0, :4].tolist() == c["parity_0_0_4"]
    assert int(par[1, 0]) == c["parity_1_0"] and int(par[-1, -1]) == c["parity_last_last"]
    assert oracle.hash(par) == c["hash_parity"]


def test_headline_properties(torch_cuda, headline, oracle):
    """Linearity, constant-polynomial fixed point and a column-slab cross-check against the oracle."""
    torch = torch_cuda
    enc, N, S = headline
    g = torch.Generator(device="cuda:0")
    g.manual_seed(2026)
    a = torch.randint(0, P, (N * S,), dtype=torch.int64, device="cuda:0", generator=g)
    b = torch.randint(0, P, (N * S,), dtype=torch.int64, device="cuda:0", generator=g)
    ab = ((a + b) % P).to(torch.int32)       # wraps into int32 bit patterns of uint32 values
    a32 = (a % (1 << 32)).to(torch.int32)
    b32 = (b % (1 << 32)).to(torch.int32)
    # int64 -> int32 conversion keeps the low 32 bits
    del a, b
    # keep 8 columns of the inputs for the oracle cross-check before encoding in place
    cols = slice(500, 508)
    a_cols = to_host(a32.view(N, S)[:, cols].contiguous())
    enc.encode(a32)
    enc.encode(b32)
    enc.encode(ab)
    s = torch.empty_like(ab)
    enc.gf_binary("add", a32, b32, s, N * S)
    torch.cuda.synchronize()
    assert torch.equal(s, ab), "encode(a+b) != encode(a)+encode(b)"
    # columns are independent transforms: the oracle on an 8-column slab must reproduce those columns
    want = oracle.encode_fast(a_cols)
    assert np.array_equal(to_host(a32.view(N, S)[:, cols].contiguous()), want)
    # constant stripe -> parity equals data
    c = torch.full((N * S,), 123456789, dtype=torch.int32, device="cuda:0")
    enc.encode(c)
    torch.cuda.synchronize()
    assert bool((c == 123456789).all())


# ------------------------------------------------------------------------------------------------
# shapes that exercise the tile kernels' bounds-checked lanes and the large-block fallback
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("log2n,S", [(16, 33), (18, 7), (19, 5), (19, 40)])
def test_large_n_with_ragged_blocks(torch_cuda, fe, oracle, log2n, S):
    """Full-depth plans (dif9/mid10/dit9 at 2^19) with block sizes that are not a multiple of the 32-word
    tile rows: lanes past the block end are masked by the buffer bounds check."""
    torch = torch_cuda
    N = 1 << log2n
    x = rand_stripe(np.random.default_rng(log2n * 100 + S), N, S)
    d = to_dev(torch, x)
    guard = torch.full((1024,), 0x5A5A5A5A, dtype=torch.int32, device="cuda:0")  # allocated right after d
    with fe.Encoder(2 * N, N, 4 * S) as enc:
        enc.encode(d)
        torch.cuda.synchronize()
    assert np.array_equal(to_host(d), oracle.encode_fast(x))
    assert bool((guard == 0x5A5A5A5A).all())


@pytest.mark.parametrize("k,m,S", [(1 << 18, 1 << 16, 6), (1 << 18, (1 << 17) + 3, 33), (200000, 50000, 5), (1 << 19, 1 << 17, 3), (400000, 100000, 4),
                                   (1 << 17, 1 << 14, 40), (1 << 16, 1 << 15, 64), (1 << 17, 3 << 17, 9)])
def test_default_plans_of_large_codes_with_other_shapes(torch_cuda, fe, oracle, k, m, S):
    """The shorter MID and the 64-word-row outer tiles are the default from k = 2^16 up for every power-of-two code shape: fewer parity blocks
    (the compact parity stripe of the passes above MID), zero extension (loads bounded by the descriptor), n = 4k (cosets), ragged rows."""
    torch = torch_cuda
    x = rand_stripe(np.random.default_rng(k % 1000 + m % 100 + S), k, S)
    lg = int(np.ceil(np.log2(k)))
    N = 1 << lg
    padded = np.zeros((N, S), dtype=np.uint32)
    padded[:k] = x
    if m > N:  # n = 4k: the cosets w_2k, w_4k, w_4k^3 of the data points (k = N here): RS.cpp:40-63 with the coset's generator for root(2N)
        coef = oracle.ntt_fast(x, inverse=True)
        gens = [oracle.gf_root(2 * N)] + [oracle.gf_pow(oracle.gf_root(4 * N), c) for c in (1, 3)]
        want = np.concatenate([oracle.ntt_fast(oracle.scale_blocks(coef, oracle.gf_inv(N), g)) for g in gens])
    else:
        fold = min(lg - (int(np.ceil(np.log2(m))) if m > 1 else 0), 4)
        want = oracle.encode_fast(padded)[:: 1 << fold][:m]
    d = to_dev(torch, x)
    out = torch.full((m * S,), 0x66666666, dtype=torch.int32, device="cuda:0")
    with fe.Encoder(k + m, k, 4 * S) as enc:
        # (these short rows keep the MID10 plan by default — the re-split is chosen from 2 KB blocks up —, so it is selected by id: 3090 / 3080 =
        #  MID9 / MID8 with the 1024-block outer tiles where the levels ask for them, 4090 = 64-word rows for 9-level outer chunks)
        for plan in (3090, 3080, 4090, 0):
            enc.set_plan(plan)
            out.fill_(0x66666666)
            enc.encode(d, out)
            torch.cuda.synchronize()
            assert np.array_equal(to_host(out).reshape(m, S), want), (plan, enc.plan())
    assert np.array_equal(to_host(d).reshape(k, S), x)


def test_the_default_plan_of_large_stripes_is_the_shorter_mid(torch_cuda, fe):
    """From 2 KB blocks and k = 2^16 up the default is the split with the shorter MID (plan.hip build_plans); it and plan 3100 give the same parity."""
    torch = torch_cuda
    for log2k, S, text in ((16, 512, "S32:dif8@8,T32:mid8@0,S32:dit8@8"), (17, 512, "S32:dif8@9,T32:mid9@0,S32:dit8@9"), (18, 513, "T64:dif9@9,T32:mid9@0,T64:dit9@9")):
        k = 1 << log2k
        d = torch.randint(0, P, (k * S,), dtype=torch.int64, device="cuda:0").to(torch.int32)
        a, b = torch.empty_like(d), torch.empty_like(d)
        with fe.Encoder(2 * k, k, 4 * S) as enc:
            assert enc.plan().startswith(text), enc.plan()
            enc.encode(d, a)
            enc.set_plan(3100)
            assert "mid10@0" in enc.plan()
            enc.encode(d, b)
            torch.cuda.synchronize()
        assert bool(torch.equal(a, b)), log2k
    with fe.Encoder(1 << 20, 1 << 19, 1024) as enc:  # 1 KB blocks: MID10 stays
        assert "mid10@0" in enc.plan()


def _big_block_case(torch, fe, oracle, N, S, expect_tiles):
    g = torch.Generator(device="cuda:0")
    g.manual_seed(7)
    d = torch.empty(N * S, dtype=torch.int32, device="cuda:0")
    for i in range(0, N * S, 1 << 26):
        m = min(1 << 26, N * S - i)
        d[i:i + m] = torch.randint(0, P, (m,), dtype=torch.int64, device="cuda:0", generator=g).to(torch.int32)
    ref_cols = [slice(0, 8), slice(S - 8, S), slice(S // 2, S // 2 + 8)]
    inputs = [to_host(d.view(N, S)[:, c].contiguous()) for c in ref_cols]
    out_default = torch.empty_like(d)
    out_reg = torch.empty_like(d)
    with fe.Encoder(2 * N, N, 4 * S) as enc:
        has_tiles = any(t in enc.plan() for t in ("T32", "T64", "S32"))
        assert has_tiles == expect_tiles, enc.plan()
        enc.encode(d, out_default)
        enc.set_plan(34)
        enc.encode(d, out_reg)
        torch.cuda.synchronize()
    assert torch.equal(out_default, out_reg)
    for c, xin in zip(ref_cols, inputs):
        assert np.array_equal(to_host(out_default.view(N, S)[:, c].contiguous()), oracle.encode_fast(xin))


def test_tile_offsets_between_2g_and_4g(torch_cuda, fe, oracle):
    """32 MiB blocks, 64-block tile: the tile spans 2^31 + 1 KiB bytes, so scalar block offsets exceed 2^31 while
    staying below the 2^32-1 bound of the buffer descriptor.  Checked against register passes and the oracle."""
    _big_block_case(torch_cuda, fe, oracle, 64, (1 << 23) + 4, expect_tiles=True)


def test_blocks_too_large_for_tile_offsets_fall_back(torch_cuda, fe, oracle):
    """64 MiB blocks: a 64-block tile would span > 2^32 bytes, so the plan must use register passes
    (64-bit addressing)."""
    _big_block_case(torch_cuda, fe, oracle, 64, (1 << 24) + 4, expect_tiles=False)


# ------------------------------------------------------------------------------------------------
# directly against the unmodified reference (prebuilt oracle/_ref travels to the GPU box)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("log2n,S,avx2", [(7, 1024, False), (12, 513, True), (15, 1024, True), (16, 256, False)])
def test_encode_equals_unmodified_reference(torch_cuda, fe, log2n, S, avx2):
    """HIP encode vs FastECC's own code (RS.cpp:41-63 call sequence on MFA_NTT, scalar and AVX2 builds)."""
    from oracle import Reference
    if not Reference.available(avx2):
        pytest.skip("oracle/_ref not prebuilt")
    torch = torch_cuda
    ref = Reference(avx2)
    N = 1 << log2n
    x = rand_stripe(np.random.default_rng(log2n + S), N, S)
    d = to_dev(torch, x)
    with fe.Encoder(2 * N, N, 4 * S) as enc:
        enc.encode(d)
        torch.cuda.synchronize()
    assert np.array_equal(to_host(d), ref.encode(x))


@pytest.mark.parametrize("log2n,S", [(8, 1024), (13, 64)])
def test_ntt_equals_unmodified_reference(torch_cuda, fe, log2n, S):
    """fastecc_ntt vs MFA_NTT and Rec_NTT of the reference, read back in logical block order."""
    from oracle import Reference
    if not Reference.available():
        pytest.skip("oracle/_ref not prebuilt")
    torch = torch_cuda
    ref = Reference()
    N = 1 << log2n
    x = rand_stripe(np.random.default_rng(log2n), N, S)
    with fe.Encoder(2 * N, N, 4 * S) as enc:
        for inverse in (False, True):
            d = to_dev(torch, x)
            enc.ntt(d, inverse)
            torch.cuda.synchronize()
            got = to_host(d)
            assert np.array_equal(got, ref.ntt(x, inverse, Reference.MFA))
            assert np.array_equal(got, ref.ntt(x, inverse, Reference.REC))


@pytest.mark.parametrize("order", [2, 4])
def test_ntt_of_order_2_and_4_equals_the_reference_codelets(torch_cuda, fe, order):
    """fastecc_ntt at N = 2, 4 against NTT2 / NTT4 (ntt.cpp:16-22, 50-62) themselves, column by column, natural
    order in and out (NTT4's closing swap included), both directions."""
    from oracle import Reference
    if not Reference.available():
        pytest.skip("oracle/_ref not prebuilt")
    ref = Reference()
    if not hasattr(ref.lib, "ref_small_ntt") or ref.lib.ref_small_ntt(np.zeros(4, np.uint32), 4, 0) != 0:
        pytest.skip("oracle/_ref predates the order-2 / order-4 entries of ref_small_ntt")
    torch = torch_cuda
    for S in (1, 33, 1024):
        x = rand_stripe(np.random.default_rng(100 * order + S), order, S)
        x[:, 0] = np.array([0, P - 1, 1, P - 2], dtype=np.uint32)[:order]
        with fe.Encoder(2 * order, order, 4 * S) as enc:
            for inverse in (False, True):
                d = to_dev(torch, x)
                enc.ntt(d, inverse)
                torch.cuda.synchronize()
                want = np.stack([ref.small_ntt(x[:, c], inverse) for c in range(S)], axis=1)
                assert np.array_equal(to_host(d), want), (order, S, inverse)


def test_check_range_counts_non_field_words(torch_cuda, fe):
    """Inputs must be < p (README.md:160-162); fastecc_check_range finds the ones that are not."""
    torch = torch_cuda
    N, S = 64, 1021  # odd width: exercises the unaligned head / tail paths
    rng = np.random.default_rng(9)
    x = rng.integers(0, P, size=(N, S), dtype=np.uint64).astype(np.uint32)
    with fe.Encoder(2 * N, N, 4 * S) as enc:
        assert enc.check_range(to_dev(torch, x)) == 0
        bad_pos = rng.choice(N * S, size=37, replace=False)
        y = x.reshape(-1).copy()
        y[bad_pos] = rng.integers(P, 1 << 32, size=37, dtype=np.uint64).astype(np.uint32)
        y[0], y[-1] = 0xFFFFFFFF, P  # first and last word
        want = int((y.astype(np.uint64) >= P).sum())
        assert enc.check_range(to_dev(torch, y)) == want
        assert enc.check_range(y.reshape(N, S), mem=fe.MEM_HOST) == want
        # a view starting 4 bytes into an allocation (not 16-byte aligned)
        z = to_dev(torch, np.concatenate([[0], y]).astype(np.uint32))
        assert enc.check_range(z.data_ptr() + 4) == want


@pytest.mark.parametrize("slabs", [2, 4, 8])
def test_column_slab_pipeline_is_bit_exact(torch_cuda, fe, oracle, slabs):
    """fastecc_set_option("slabs", H): H column slabs on internal streams, staggered by one pass."""
    torch = torch_cuda
    for log2n, S in [(12, 256), (16, 512), (17, 96)]:  # 96 is not divisible by 32*H for H = 4, 8: falls back to one stream
        N = 1 << log2n
        x = rand_stripe(np.random.default_rng(slabs * 1000 + log2n), N, S)
        want = oracle.encode_fast(x)
        d = to_dev(torch, x)
        out = torch.empty_like(d)
        st = torch.cuda.Stream()
        with fe.Encoder(2 * N, N, 4 * S) as enc:
            enc.set_option("slabs", slabs)
            with torch.cuda.stream(st):
                enc.encode(d, out, stream=st.cuda_stream)   # out of place
                enc.encode(d, stream=st.cuda_stream)        # and in place, queued behind it on the same stream
            st.synchronize()
        assert np.array_equal(to_host(out), want), (slabs, log2n, S)
        assert np.array_equal(to_host(d), want), (slabs, log2n, S)


@pytest.mark.parametrize("log2n,S,pitch", [(10, 513, 544), (16, 1025, 1056), (12, 7, 32), (9, 64, 65)])
def test_row_pitch(torch_cuda, fe, oracle, log2n, S, pitch):
    """fastecc_set_option("row_pitch_words", L): stripes are [k][L] words, only the first S of a row are data."""
    torch = torch_cuda
    N = 1 << log2n
    x = rand_stripe(np.random.default_rng(log2n + pitch), N, S)
    want = oracle.encode_fast(x)
    padded = np.full((N, pitch), 0xDEADBEEF, dtype=np.uint32)
    padded[:, :S] = x
    d = to_dev(torch, padded)
    out = torch.full_like(d, 0x0BADF00D)
    with fe.Encoder(2 * N, N, 4 * S) as enc:
        enc.set_option("row_pitch_words", pitch)
        enc.encode(d, out)
        torch.cuda.synchronize()
        got = to_host(out).reshape(N, pitch)
        assert np.array_equal(got[:, :S], want)
        assert (got[:, S:] == 0x0BADF00D).all(), "padding words of the output must not be written"
        enc.encode(d)  # in place
        torch.cuda.synchronize()
        got = to_host(d).reshape(N, pitch)
        assert np.array_equal(got[:, :S], want) and (got[:, S:] == 0xDEADBEEF).all()
        with pytest.raises(fe.FastEccError):
            enc.ntt(d)
        enc.set_option("row_pitch_words", 0)
        c = to_dev(torch, x)
        enc.encode(c)
        torch.cuda.synchronize()
        assert np.array_equal(to_host(c), want)


@pytest.mark.parametrize("log2n,S,plan,slabs", [(4, 64, 0, 8), (11, 256, 0, 8), (11, 256, 0, 2), (12, 96, 0, 4), (12, 100, 0, 8), (12, 256, 52, 8), (14, 1024, 0, 8)])
def test_pinned_host_stripes_through_the_slab_pipeline(torch_cuda, fe, oracle, log2n, S, plan, slabs):
    """FASTECC_MEM_HOST_PINNED: strided copy-engine uploads, kernels and downloads per column slab; same parity."""
    torch = torch_cuda
    N = 1 << log2n
    x = rand_stripe(np.random.default_rng(log2n + S), N, S)
    want = oracle.encode_fast(x)
    hdata = torch.from_numpy(x.view(np.int32).copy()).pin_memory()
    hpar = torch.zeros(N * S, dtype=torch.int32).pin_memory()
    with fe.Encoder(2 * N, N, 4 * S) as enc:
        if plan:
            enc.set_plan(plan)
        enc.set_option("host_slabs", slabs)
        enc.encode(hdata.data_ptr(), hpar.data_ptr(), stream=torch.cuda.current_stream().cuda_stream, mem=fe.MEM_HOST_PINNED)
        torch.cuda.synchronize()
        assert np.array_equal(hpar.numpy().view(np.uint32).reshape(N, S), want), enc.plan()
        assert np.array_equal(hdata.numpy().view(np.uint32).reshape(N, S), x)
        with pytest.raises(fe.FastEccError):
            enc.ntt(hdata.data_ptr(), mem=fe.MEM_HOST_PINNED)   # only fastecc_encode knows this kind


@pytest.mark.parametrize("S,tag", [(4096, "SW32:"), (8192, "SW4x32:"), (16384, "SW8x32:")])
def test_windowed_tiles_for_blocks_spanning_4_to_32_gib(torch_cuda, fe, oracle, S, tag):
    """k = 2^18 blocks of 16 / 32 / 64 KiB: the outer 8-level tiles span 2^32 / 2^33 / 2^34 bytes and use 2 / 4 / 8
    address windows per tile (TileArgs::wide).  Checked against the register-pass plan on the device and against the
    oracle on sampled columns."""
    torch = torch_cuda
    N = 1 << 18
    free, _ = torch.cuda.mem_get_info()
    if free < 3.3 * N * S * 4 + (2 << 30):
        pytest.skip("needs %d GiB of HBM" % (3.3 * N * S * 4 / 2**30))
    g = torch.Generator(device="cuda:0").manual_seed(23)
    data = torch.empty(N * S, dtype=torch.int32, device="cuda:0")
    for i in range(0, N * S, 1 << 28):
        data[i:i + (1 << 28)] = torch.randint(0, P, (min(1 << 28, N * S - i),), dtype=torch.int64, device="cuda:0", generator=g).to(torch.int32)
    a, b = torch.empty_like(data), torch.empty_like(data)
    with fe.Encoder(2 * N, N, 4 * S) as enc:
        assert tag in enc.plan(), enc.plan()
        enc.encode(data, a)
        enc.set_plan(51)
        enc.encode(data, b)
        torch.cuda.synchronize()
    assert bool((a == b).all())
    d2, a2 = data.view(N, S), a.view(N, S)
    for c in (0, 31, 32, 4097 % S, S - 1):
        x = d2[:, c:c + 1].contiguous().cpu().numpy().view(np.uint32)
        assert np.array_equal(a2[:, c:c + 1].contiguous().cpu().numpy().view(np.uint32), oracle.encode_fast(x)), c


@pytest.mark.parametrize("log2n,S,count,plan", [(1, 8, 5, 0), (4, 70, 3, 0), (7, 1024, 16, 0), (7, 1024, 7, 51), (10, 64, 9, 0), (11, 100, 4, 0),
                                              (13, 32, 3, 0), (13, 32, 2, 1090), (12, 48, 5, 32)])
def test_batched_stripes_equal_individual_encodes(torch_cuda, fe, oracle, log2n, S, count, plan):
    """fastecc_encode_batch: `count` stripes back to back, one launch per pass, same parity as stripe-by-stripe."""
    torch = torch_cuda
    N = 1 << log2n
    x = rand_stripe(np.random.default_rng(log2n * 100 + count), N * count, S)
    want = np.concatenate([oracle.encode_fast(x[b * N:(b + 1) * N]) for b in range(count)])
    with fe.Encoder(2 * N, N, 4 * S) as enc:
        if plan:
            enc.set_plan(plan)
        d = to_dev(torch, x)
        out = torch.empty_like(d)
        enc.encode_batch(d, out, count)
        torch.cuda.synchronize()
        assert np.array_equal(to_host(out).reshape(N * count, S), want), enc.plan()
        enc.encode_batch(d, None, count)   # in place
        torch.cuda.synchronize()
        assert np.array_equal(to_host(d).reshape(N * count, S), want)
    with fe.Encoder(N + max(1, N // 2), N, 4 * S) as enc:
        if N >= 2:
            with pytest.raises(fe.FastEccError) as ei:
                enc.encode_batch(d, out, count)
            assert ei.value.code == fe.E_UNSUPPORTED


def test_first_call_does_not_synchronise_and_tables_follow_the_streams(torch_cuda, fe, oracle):
    """The twiddle tables of a context are written by a kernel on the stream of the first call, with no host synchronisation
    (include/fastecc.h: DEVICE calls never synchronise): (a) a first call on a side stream followed at once by a call on another
    stream — which must wait for that build on the device — gives the right parity both times; (b) the first call of a context can be
    captured into a graph and replayed."""
    torch = torch_cuda
    N, S = 1 << 12, 256
    x = np.random.default_rng(42).integers(0, P, size=(N, S), dtype=np.uint64).astype(np.uint32)
    want = oracle.encode_fast(x)
    d = torch.from_numpy(x.view(np.int32)).to("cuda:0")
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    for _ in range(3):  # fresh contexts: every round builds its tables again
        with fe.Encoder(2 * N, N, 4 * S) as enc:
            o1, o2, t0, t = torch.empty_like(d), torch.empty_like(d), d.clone(), d.clone()
            torch.cuda.synchronize()                     # (the clones ran on torch's stream: the library's streams below do not wait for it)
            enc.encode(d, o1, stream=s1.cuda_stream)
            enc.encode(d, o2, stream=s2.cuda_stream)
            enc.ntt(t0, stream=s2.cuda_stream)           # another table kind, first built on s2 ...
            enc.ntt(t, stream=s1.cuda_stream)            # ... and used on s1 right away
            torch.cuda.synchronize()
            assert np.array_equal(o1.cpu().numpy().view(np.uint32).reshape(N, S), want)
            assert np.array_equal(o2.cpu().numpy().view(np.uint32).reshape(N, S), want)
            assert np.array_equal(t.cpu().numpy().view(np.uint32).reshape(N, S), oracle.ntt_fast(x))
    with fe.Encoder(2 * N, N, 4 * S) as enc:
        out = torch.zeros_like(d)
        g = torch.cuda.CUDAGraph()
        cap = torch.cuda.Stream()
        try:
            with torch.cuda.graph(g, stream=cap, capture_error_mode="relaxed"):
                enc.encode(d, out, stream=torch.cuda.current_stream().cuda_stream)
        except Exception as e:  # noqa: BLE001
            pytest.fail("the first fastecc_encode of a context could not be captured: %r" % (e,))
        torch.cuda.synchronize()
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy().view(np.uint32).reshape(N, S), want)
    # (c) the captured call is the context's FIRST: its tables must exist before any replay (they are built outside the capture), so an
    # eager call on another stream, and one on the capture stream itself, are right before the graph has ever run; then the replay is too
    with fe.Encoder(2 * N, N, 4 * S) as enc:
        out, o1, o2, t1 = torch.zeros_like(d), torch.zeros_like(d), torch.zeros_like(d), d.clone()
        g = torch.cuda.CUDAGraph()
        cap = torch.cuda.Stream()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=cap, capture_error_mode="relaxed"):
            enc.encode(d, out, stream=torch.cuda.current_stream().cuda_stream)
        enc.encode(d, o1, stream=s1.cuda_stream)      # another stream, no replay yet
        enc.encode(d, o2, stream=cap.cuda_stream)     # the stream the capture ran on, no replay yet
        enc.ntt(t1, stream=s2.cuda_stream)            # a table kind the capture never touched: the ordinary lazy build
        torch.cuda.synchronize()
        assert np.array_equal(o1.cpu().numpy().view(np.uint32).reshape(N, S), want)
        assert np.array_equal(o2.cpu().numpy().view(np.uint32).reshape(N, S), want)
        assert np.array_equal(t1.cpu().numpy().view(np.uint32).reshape(N, S), oracle.ntt_fast(x))
        assert not out.any()                           # the captured call itself has not run
        for _ in range(2):
            out.zero_()
            g.replay()
            torch.cuda.synchronize()
            assert np.array_equal(out.cpu().numpy().view(np.uint32).reshape(N, S), want)
