"""GPU parity tests (-m gpu) for FASTECC_FIELD_GF_P61_SQUARED, the 64-bit-field configuration
(BASELINE.json configs[4]: 64 KB blocks over GF((2^61-1)^2)).

PARITY UNPINNED upstream: the reference has no code for this field.  The checker is our oracle
(oracle/fastecc_oracle_p61.c), itself checked against the independent big-integer vectors of
tests/golden/golden_p61.json; every comparison here is bit-exact."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

P61 = (1 << 61) - 1
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    return torch


@pytest.fixture(scope="module")
def fe(hip_lib):
    import fastecc_amd
    return fastecc_amd


@pytest.fixture(scope="module")
def orc61():
    import oracle
    return oracle.OracleP61()


def to_dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to("cuda:0")


def to_host(t):
    return t.cpu().numpy().view(np.uint64)


def rand_stripe(rng, N, elems):
    x = rng.integers(0, P61, size=(N, 2 * elems), dtype=np.uint64)
    flat = x.reshape(-1)
    edge = [0, 1, P61 - 1, P61 - 2, 1 << 31, (1 << 31) - 1, 1 << 60, (1 << 60) + (1 << 30), (1 << 32) - 1]
    flat[: min(len(edge), flat.size)] = edge[: flat.size]
    return x


def encoder(fe, N, elems):
    return fe.Encoder(2 * N, N, 16 * elems, field=fe.FIELD_GF_P61_SQUARED)


def test_golden_vectors(torch_cuda, fe):
    doc = json.load(open(os.path.join(HERE, "golden", "golden_p61.json")))
    for case in doc["cases"]:
        N, elems = case["N"], case["elems"]
        x = np.array([int(w) for w in case["data"]], dtype=np.uint64).reshape(N, 2 * elems)
        want = np.array([int(w) for w in case["parity"]], dtype=np.uint64).reshape(N, 2 * elems)
        with encoder(fe, N, elems) as enc:
            d = to_dev(torch_cuda, x)
            out = torch_cuda.empty_like(d)
            enc.encode(d, out)
            assert (to_host(out) == want).all(), (N, elems)
            assert (to_host(d) == x).all()  # out of place leaves the data alone


@pytest.mark.parametrize("logn", [1, 2, 3, 4, 5, 6, 7, 9, 10, 12])
@pytest.mark.parametrize("elems", [1, 3, 64, 100])
def test_encode_matches_oracle(torch_cuda, fe, orc61, logn, elems):
    N = 1 << logn
    if N * elems > (1 << 17):
        pytest.skip("oracle time")
    x = rand_stripe(np.random.default_rng(1000 * logn + elems), N, elems)
    want = orc61.encode(x)
    with encoder(fe, N, elems) as enc:
        d = to_dev(torch_cuda, x)
        enc.encode(d)  # in place, the reference's behaviour
        got = to_host(d)
    assert (got < P61).all(), "parity words must be canonical"
    assert (got == want).all()


@pytest.mark.parametrize("plan", [1, 2, 3, 4, 0, 12, 13, 14, 24, 23])
@pytest.mark.parametrize("logn", [3, 6, 7, 8, 11, 12, 13, 14])
def test_every_plan(torch_cuda, fe, orc61, plan, logn):
    """register passes (1..5 levels), LDS tiles with a 64 KiB (0, 1x) and a 128 KiB (2x) exchange buffer: bit-identical;
    the sizes cover MID tiles (6, 7), one and two tile chunks around a MID (12, 13, 14) and mixed tile / register plans"""
    N, elems = 1 << logn, (37 if logn < 12 else 70)
    x = rand_stripe(np.random.default_rng(plan * 100 + logn), N, elems)
    want = orc61.encode(x)
    with encoder(fe, N, elems) as enc:
        enc.set_plan(plan)
        d = to_dev(torch_cuda, x)
        out = torch_cuda.empty_like(d)
        enc.encode(d, out)
        assert (to_host(out) == want).all(), enc.plan()


@pytest.mark.parametrize("logn", [1, 4, 6, 9, 13])
def test_ntt_matches_oracle_and_inverts(torch_cuda, fe, orc61, logn):
    N, elems = 1 << logn, 5
    x = rand_stripe(np.random.default_rng(logn), N, elems)
    with encoder(fe, N, elems) as enc:
        for inverse in (False, True):
            d = to_dev(torch_cuda, x)
            enc.ntt(d, inverse=inverse)
            assert (to_host(d) == orc61.ntt(x, inverse)).all(), (logn, inverse)
        # inverse(forward(x)) = N * x  (both transforms are unscaled, ntt.cpp:451-483)
        d = to_dev(torch_cuda, x)
        enc.ntt(d)
        enc.ntt(d, inverse=True)
        got = to_host(d).astype(object)
        assert (got == (x.astype(object) * N) % P61).all()


def test_host_stripe_and_block_pointers(torch_cuda, fe, orc61):
    N, elems = 64, 9
    x = rand_stripe(np.random.default_rng(5), N, elems)
    want = orc61.encode(x)
    with encoder(fe, N, elems) as enc:
        out = np.empty_like(x)
        enc.encode_host(x, out)
        assert (out == want).all()
        blocks = [np.ascontiguousarray(x[i]).copy() for i in range(N)]
        enc.encode_blocks([b.ctypes.data for b in blocks])
        assert (np.stack(blocks) == want).all()


def test_check_range(torch_cuda, fe):
    N, elems = 32, 16
    x = rand_stripe(np.random.default_rng(6), N, elems)
    with encoder(fe, N, elems) as enc:
        d = to_dev(torch_cuda, x)
        assert enc.check_range(d) == 0
        x[3, 5] = P61
        x[31, 31] = (1 << 64) - 1
        assert enc.check_range(to_dev(torch_cuda, x)) == 2


def test_unsupported_entry_points(torch_cuda, fe):
    with encoder(fe, 4, 4) as enc:
        d = torch_cuda.zeros(4 * 8, dtype=torch_cuda.int64, device="cuda:0")
        with pytest.raises(fe.FastEccError) as ei:
            enc.scale_blocks(d, 1, 1)
        assert ei.value.code == fe.E_UNSUPPORTED
        with pytest.raises(fe.FastEccError):
            enc.set_option("slabs", 2)
    with pytest.raises(fe.FastEccError) as ei:
        fe.Encoder(8, 4, 24, field=fe.FIELD_GF_P61_SQUARED)  # block_bytes % 16
    assert ei.value.code == fe.E_INVAL


def sample_columns_check(torch, orc61, data_dev, parity_dev, N, elems, cols):
    """Columns are independent transforms: the oracle re-encodes a few of them exactly."""
    d = data_dev.view(N, 2 * elems)
    p = parity_dev.view(N, 2 * elems)
    for c in cols:
        x = d[:, 2 * c:2 * c + 2].contiguous().cpu().numpy().view(np.uint64)
        got = p[:, 2 * c:2 * c + 2].contiguous().cpu().numpy().view(np.uint64)
        assert (got == orc61.encode(x)).all(), c


def test_linearity_and_sampled_columns_2_16(torch_cuda, fe, orc61):
    torch = torch_cuda
    N, elems = 1 << 16, 128
    g = torch.Generator(device="cuda:0").manual_seed(7)
    a = torch.randint(0, P61, (N * 2 * elems,), dtype=torch.int64, device="cuda:0", generator=g)
    b = torch.randint(0, P61, (N * 2 * elems,), dtype=torch.int64, device="cuda:0", generator=g)
    s = a + b
    s = torch.where(s >= P61, s - P61, s)
    with encoder(fe, N, elems) as enc:
        ea, eb, es = torch.empty_like(a), torch.empty_like(b), torch.empty_like(s)
        enc.encode(a, ea)
        enc.encode(b, eb)
        enc.encode(s, es)
        t = ea + eb
        t = torch.where(t >= P61, t - P61, t)
        assert bool((t == es).all())
        sample_columns_check(torch, orc61, a, ea, N, elems, [0, 63, 64, 127])


def test_headline_size_sampled_columns(torch_cuda, fe, orc61):
    """(n,k) = (2^20, 2^19), 64 KB blocks: 32 GiB of data, encoded in place; four columns re-encoded by the oracle."""
    torch = torch_cuda
    N, elems = 1 << 19, 4096
    free, _ = torch.cuda.mem_get_info()
    if free < 70 * (1 << 30):
        pytest.skip("needs 64 GiB of HBM")
    g = torch.Generator(device="cuda:0").manual_seed(19)
    data = torch.randint(0, P61, (N * 2 * elems,), dtype=torch.int64, device="cuda:0", generator=g)
    cols = [0, 1, 2047, 4095]
    keep = {c: data.view(N, 2 * elems)[:, 2 * c:2 * c + 2].contiguous().cpu().numpy().view(np.uint64) for c in cols}
    with encoder(fe, N, elems) as enc:
        assert enc.check_range(data) == 0
        enc.encode(data)
        torch.cuda.synchronize()
    par = data.view(N, 2 * elems)
    for c in cols:
        got = par[:, 2 * c:2 * c + 2].contiguous().cpu().numpy().view(np.uint64)
        assert (got == orc61.encode(keep[c])).all(), c


# ------------------------------------------------------------------------------------------------
# erasure decoder over this field (gf61_decode.hip): round trips and the oracle's O(N^2) Lagrange decoder
# ------------------------------------------------------------------------------------------------
def loss_pattern(rng, N, kind):
    dp, pp = np.ones(N, np.uint8), np.ones(N, np.uint8)
    if kind == "one":
        dp[rng.integers(N)] = 0
    elif kind == "all_data":
        dp[:] = 0
    elif kind == "random_max":
        lost = rng.permutation(2 * N)[:N]
        dp[lost[lost < N]] = 0
        pp[lost[lost >= N] - N] = 0
    elif kind == "quarter":
        lost = rng.permutation(2 * N)[: max(1, N // 2)]
        dp[lost[lost < N]] = 0
        pp[lost[lost >= N] - N] = 0
    elif kind == "burst":
        dp[N // 4: N // 4 + max(1, N // 3)] = 0
    return dp, pp


@pytest.mark.parametrize("logn", [1, 2, 3, 4, 5, 6, 8, 10, 12])
@pytest.mark.parametrize("kind", ["one", "all_data", "random_max", "quarter", "burst"])
def test_decode_round_trip_and_lagrange(torch_cuda, fe, orc61, logn, kind):
    torch = torch_cuda
    N, elems = 1 << logn, 5 if logn % 2 else 70
    rng = np.random.default_rng(logn * 10 + len(kind))
    x = rand_stripe(rng, N, elems)
    par = orc61.encode(x)
    dp, pp = loss_pattern(rng, N, kind)
    damaged, dpar = x.copy(), par.copy()
    damaged[dp == 0] = np.uint64(0xFFFFFFFFFFFFFFFF)   # erased blocks hold garbage (not even field elements)
    dpar[pp == 0] = np.uint64(0xDEADBEEFDEADBEEF)
    with encoder(fe, N, elems) as enc:
        enc.decode_prepare(dp, pp)
        d, q = to_dev(torch, damaged), to_dev(torch, dpar)
        enc.decode(d, q)
        torch.cuda.synchronize()
        assert (to_host(d).reshape(x.shape) == x).all()
        assert (to_host(q).reshape(par.shape) == dpar).all()  # decode leaves the parity alone
        enc.repair(to_dev(torch, damaged), q)
        torch.cuda.synchronize()
        assert (to_host(q).reshape(par.shape) == par).all()   # repair brings the lost parity blocks back too
    if N <= 64:
        assert (orc61.decode(damaged, dpar, dp, pp) == x).all()


def test_decode_patterns_change_and_errors(torch_cuda, fe, orc61):
    torch = torch_cuda
    N, elems = 128, 9
    rng = np.random.default_rng(99)
    x = rand_stripe(rng, N, elems)
    par = orc61.encode(x)
    with encoder(fe, N, elems) as enc:
        with pytest.raises(fe.FastEccError):
            enc.decode(to_dev(torch, x), to_dev(torch, par))  # no pattern set
        for kind in ("one", "random_max", "burst", "one"):
            dp, pp = loss_pattern(rng, N, kind)
            damaged = x.copy()
            damaged[dp == 0] = 7
            enc.decode_prepare(dp, pp)
            d = to_dev(torch, damaged)
            enc.decode(d, to_dev(torch, par))
            torch.cuda.synchronize()
            assert (to_host(d).reshape(x.shape) == x).all(), kind
        dp, pp = np.zeros(N, np.uint8), np.ones(N, np.uint8)
        pp[0] = 0
        with pytest.raises(fe.FastEccError) as ei:  # N + 1 losses
            enc.decode_prepare(dp, pp)
        assert ei.value.code == fe.E_INVAL
        # the refused pattern changed nothing: the last accepted one still decodes — here from stripes in host memory
        host = damaged.copy()
        enc.decode(host, par, mem=fe.MEM_HOST)
        assert (host == x).all()


def test_decode_at_2_16_blocks(torch_cuda, fe):
    """(2^17, 2^16) x 1 KiB blocks, half of the codeword lost: encode -> erase -> repair == original (no oracle at this size)."""
    torch = torch_cuda
    N, elems = 1 << 16, 64
    g = torch.Generator(device="cuda:0")
    g.manual_seed(5)
    x = torch.randint(0, P61, (N * 2 * elems,), dtype=torch.int64, device="cuda:0", generator=g)
    par = torch.empty_like(x)
    rng = np.random.default_rng(3)
    lost = rng.permutation(2 * N)[:N]
    dp, pp = np.ones(N, np.uint8), np.ones(N, np.uint8)
    dp[lost[lost < N]] = 0
    pp[lost[lost >= N] - N] = 0
    with encoder(fe, N, elems) as enc:
        enc.encode(x, par)
        d, q = x.clone(), par.clone()
        d.view(N, 2 * elems)[torch.from_numpy(dp == 0).to("cuda:0")] = -1
        q.view(N, 2 * elems)[torch.from_numpy(pp == 0).to("cuda:0")] = -1
        enc.decode_prepare(dp, pp)
        enc.repair(d, q)
        torch.cuda.synchronize()
        assert torch.equal(d, x) and torch.equal(q, par)


@pytest.mark.parametrize("logn,elems", [(6, 3), (7, 70), (9, 64), (12, 70), (13, 20), (14, 64), (18, 2)])
def test_decode_transform_is_folded(torch_cuda, fe, logn, elems):
    """Only the k data positions of the decoder's 2k-point transform are read, so it runs as big DIF passes, one folding MID tile
    (y[j] = x[2j] + x[2j+1] between the halves of a 7-level MID) and the DIT passes of the size-k path: the profile shows that kernel,
    and the decode is the same as before (round trip).  Where the plan has them, the first DIF tile also gathers (codeword position -> data or parity
    block, times l(w^u), erased positions never read) and the last DIT tile scatters (only the rebuilt blocks are written, times their factor)."""
    torch = torch_cuda
    N = 1 << logn
    g = torch.Generator(device="cuda:0")
    g.manual_seed(logn)
    x = torch.randint(0, P61, (N * 2 * elems,), dtype=torch.int64, device="cuda:0", generator=g)
    par = torch.empty_like(x)
    rng = np.random.default_rng(logn)
    lost = rng.permutation(2 * N)[: N - 1]
    dp, pp = np.ones(N, np.uint8), np.ones(N, np.uint8)
    dp[lost[lost < N]] = 0
    pp[lost[lost >= N] - N] = 0
    with encoder(fe, N, elems) as enc:
        enc.encode(x, par)
        d, q = x.clone(), par.clone()
        d.view(N, 2 * elems)[torch.from_numpy(dp == 0).to("cuda:0")] = -1
        q.view(N, 2 * elems)[torch.from_numpy(pp == 0).to("cuda:0")] = -1
        enc.decode_prepare(dp, pp)
        enc.profile(True)
        enc.profile_reset()
        enc.decode(d, q)
        torch.cuda.synchronize()
        prof = enc.profile_read()
        enc.profile(False)
        assert prof.get("p61_tile_mid7_fold", (0, 0, 0))[1] == 1, prof
        if logn in (12, 13, 18):
            # the transform starts with a DIF tile and ends with a DIT tile there: the gather and the scatter ride in them (6- and 7-level forms)
            assert any(name.endswith("_gather") for name in prof) and any(name.endswith("_scatter") for name in prof), prof
        assert torch.equal(d, x)
        d2 = x.clone()
        d2.view(N, 2 * elems)[torch.from_numpy(dp == 0).to("cuda:0")] = -1
        enc.profile(True)
        enc.profile_reset()
        enc.repair(d2, q)
        torch.cuda.synchronize()
        prof = enc.profile_read()
        enc.profile(False)
        assert torch.equal(d2, x) and torch.equal(q, par)
        if logn in (12, 13, 18):
            # data AND parity lost: ONE transform over all 2k positions (no fold, no second encode), gather and scatter in its end tiles
            assert "p61_tile_mid7_fold" not in prof and any(name.endswith("_gather") for name in prof) and any(name.endswith("_scatter") for name in prof), prof
            assert not any(name.startswith("p61_tile_mid6") for name in prof), prof  # (that would be the encoder's MID: a re-encode)


@pytest.mark.parametrize("N,elems", [(2, 3), (16, 70), (1024, 9), (1 << 14, 4)])
def test_few_losses_take_the_direct_path(torch_cuda, fe, orc61, N, elems):
    """Up to 32 lost blocks: recomputed straight from k of the survivors — the data and as many parity blocks as data blocks are lost — (no locator
    tree, no transform); same bits as the transform
    path (decode_direct_max = 0), the original stripes and — small N — the oracle's Lagrange decoder; decode and repair."""
    torch = torch_cuda
    rng = np.random.default_rng(N * 3 + elems)
    x = rand_stripe(rng, N, elems)
    par = orc61.encode(x)
    with encoder(fe, N, elems) as enc:
        for e in (1, 2, 3, 7, 16, 17, 32, 33):
            if e > N:
                continue
            lost = np.unique(np.r_[int(rng.integers(0, N)), rng.permutation(2 * N)[: e - 1]])
            dp, pp = np.ones(N, np.uint8), np.ones(N, np.uint8)
            dp[lost[lost < N]] = 0
            pp[lost[lost >= N] - N] = 0
            bad_x, bad_p = x.copy(), par.copy()
            bad_x[dp == 0] = 11
            bad_p[pp == 0] = 13
            for direct_max in (32, 0):
                enc.set_option("decode_direct_max", direct_max)
                enc.decode_prepare(dp, pp)
                d, q = to_dev(torch, bad_x), to_dev(torch, bad_p)
                enc.decode(d, q)
                torch.cuda.synchronize()
                assert (to_host(d).reshape(x.shape) == x).all(), (e, direct_max)
                assert (to_host(q).reshape(x.shape) == bad_p).all(), (e, direct_max)
                enc.repair(d, q)
                torch.cuda.synchronize()
                assert (to_host(q).reshape(x.shape) == par).all(), (e, direct_max)
                hx, hp = bad_x.copy(), bad_p.copy()
                enc.repair(hx, hp, mem=fe.MEM_HOST)
                assert (hx == x).all() and (hp == par).all(), (e, direct_max)
            if N <= 64:
                assert (orc61.decode(bad_x, bad_p, dp, pp) == x).all()
        enc.set_option("decode_direct_max", 32)


@pytest.mark.parametrize("k,m,elems", [(100, 30, 8), (256, 64, 5), (1000, 1000, 4), (300, 512, 6), (4096, 256, 16), (5, 1, 3), (2048, 2048 // 16, 7)])
def test_other_n_k_over_the_64_bit_field(torch_cuda, fe, orc61, k, m, elems):
    """(n,k) other than (2N,N) over GF((2^61-1)^2): the code definition of the 32-bit field's rules (RS.md:23-33) — the k data blocks
    zero-extended to N = 2^ceil(log2 k), parity block j = block j * 2^fold of the (2N,N) parity, fold = min(log2 N - ceil(log2 m), 4) —
    checked against the oracle on the padded stripe; then n - k random losses over data and parity are decoded and repaired."""
    torch = torch_cuda
    rng = np.random.default_rng(k * 7 + m)
    lg = max(1, int(np.ceil(np.log2(k))))
    N = 1 << lg
    stride = 1 << min(lg - int(np.ceil(np.log2(m))) if m > 1 else lg, 4)
    x = rand_stripe(rng, k, elems)
    xpad = np.zeros((N, 2 * elems), dtype=np.uint64)
    xpad[:k] = x
    want = orc61.encode(xpad)[::stride][:m]
    with fe.Encoder(k + m, k, 16 * elems, field=fe.FIELD_GF_P61_SQUARED) as enc:
        dx = to_dev(torch, x)
        out = torch.full((m * 2 * elems,), 7, dtype=torch.int64, device="cuda:0")
        enc.encode(dx, out)
        torch.cuda.synchronize()
        assert (to_host(out).reshape(m, 2 * elems) == want).all()
        assert (to_host(dx).reshape(x.shape) == x).all()
        host_out = np.empty_like(want)
        enc.encode(x, host_out, mem=fe.MEM_HOST)
        assert (host_out == want).all()
        if m <= k:
            enc.encode(dx)  # in place
            got = to_host(dx).reshape(x.shape)
            assert (got[:m] == want).all() and (got[m:] == x[m:]).all()
        for nlost in sorted({1, min(m, 5), m}):
            lost = rng.permutation(k + m)[:nlost]
            dp, pp = np.ones(k, np.uint8), np.ones(m, np.uint8)
            dp[lost[lost < k]] = 0
            pp[lost[lost >= k] - k] = 0
            bad_x, bad_p = x.copy(), want.copy()
            bad_x[dp == 0] = 11
            bad_p[pp == 0] = 13
            enc.decode_prepare(dp, pp)
            d, q = to_dev(torch, bad_x), to_dev(torch, bad_p)
            enc.decode(d, q)
            torch.cuda.synchronize()
            assert (to_host(d).reshape(x.shape) == x).all(), nlost
            assert (to_host(q).reshape(want.shape) == bad_p).all(), nlost
            enc.repair(d, q)
            torch.cuda.synchronize()
            assert (to_host(d).reshape(x.shape) == x).all() and (to_host(q).reshape(want.shape) == want).all(), nlost
        # one loss too many
        if m < k:
            dp, pp = np.ones(k, np.uint8), np.ones(m, np.uint8)
            dp[: m + 1] = 0
            with pytest.raises(fe.FastEccError):
                enc.decode_prepare(dp, pp)


def p61_coset_generators(orc61, N, e):
    """w_2N; w_4N, w_4N^3; w_8N, w_8N^3, w_8N^5, w_8N^7 — the nesting order of include/fastecc.h."""
    gens = []
    for j in range(1, e + 1):
        w = orc61.root(N << j)
        gens += [orc61.cpow(w, c) for c in range(1, 1 << j, 2)]
    return gens


def p61_oracle_coset_parity(orc61, x, e):
    """RS.cpp:40-63 with each coset generator in place of root(2N): iNTT, block i *= g^i / N, NTT."""
    N = x.shape[0]
    coef = orc61.ntt(x, inverse=True)
    inv_n = orc61.cinv((N % P61, 0))
    return np.concatenate([orc61.ntt(orc61.scale_blocks(coef, inv_n, g)) for g in p61_coset_generators(orc61, N, e)])


def test_multi_coset_golden_vectors(torch_cuda, fe, orc61):
    """n = 4k / 8k over the 64-bit field against the independent big-integer vectors (tests/golden/make_golden_p61.py: the parity blocks straight
    from the definition f(g w_k^j)), and the oracle's composition against the same vectors."""
    doc = json.load(open(os.path.join(HERE, "golden", "golden_p61.json")))
    for case in doc["coset_cases"]:
        N, elems, e = case["N"], case["elems"], case["e"]
        rows = ((1 << e) - 1) * N
        x = np.array([int(w) for w in case["data"]], dtype=np.uint64).reshape(N, 2 * elems)
        want = np.array([int(w) for w in case["parity"]], dtype=np.uint64).reshape(rows, 2 * elems)
        assert (p61_oracle_coset_parity(orc61, x, e) == want).all(), (N, e)
        with fe.Encoder(N << e, N, 16 * elems, field=fe.FIELD_GF_P61_SQUARED) as enc:
            d = to_dev(torch_cuda, x)
            out = torch_cuda.empty(rows * 2 * elems, dtype=torch_cuda.int64, device="cuda:0")
            enc.encode(d, out)
            assert (to_host(out).reshape(rows, 2 * elems) == want).all(), (N, elems, e)


@pytest.mark.parametrize("logn,elems", [(6, 1), (7, 70), (9, 64), (11, 66), (12, 2), (13, 20)])
def test_decode_of_the_n_equals_4k_code_is_folded(torch_cuda, fe, orc61, logn, elems):
    """n = 4k: only the k data positions (the multiples of 4) of the decoder's 4k-point transform are wanted, so it runs as the big path's DIF
    passes, ONE folding MID tile (four consecutive positions into one, the second half of the quarter-size path's 5-level MID) and the DIT passes
    of the size-k path — the profile shows that kernel — and where the plan starts and ends with tiles the gather (through the position map) and
    the scatter ride in them.  Round trip with the LAST tolerable loss (3k blocks); at k = 64 the codeword is the independent big-integer vector
    of tests/golden/golden_p61.json, so the decoder's output is pinned to it, not only to this library's encoder."""
    torch = torch_cuda
    N, e = 1 << logn, 2
    rows = 3 * N
    if (logn, elems) == (6, 1):
        case = [c for c in json.load(open(os.path.join(HERE, "golden", "golden_p61.json")))["coset_cases"] if c["N"] == 64 and c["e"] == 2][0]
        x = np.array([int(w) for w in case["data"]], dtype=np.uint64).reshape(N, 2 * elems)
        want = np.array([int(w) for w in case["parity"]], dtype=np.uint64).reshape(rows, 2 * elems)
    else:
        x = rand_stripe(np.random.default_rng(4000 + logn), N, elems)
        want = p61_oracle_coset_parity(orc61, x, e)
    with fe.Encoder(N << e, N, 16 * elems, field=fe.FIELD_GF_P61_SQUARED) as enc:
        out = torch.empty(rows * 2 * elems, dtype=torch.int64, device="cuda:0")
        enc.encode(to_dev(torch, x), out)
        assert (to_host(out).reshape(rows, 2 * elems) == want).all()
        rng = np.random.default_rng(logn)
        for nlost in (3 * N, N):
            lost = rng.permutation(4 * N)[:nlost]
            dp, pp = np.ones(N, np.uint8), np.ones(rows, np.uint8)
            dp[lost[lost < N]] = 0
            pp[lost[lost >= N] - N] = 0
            if not (dp == 0).any():
                dp[0] = 0
                pp[np.flatnonzero(pp == 0)[0]] = 1
            bad_x, bad_p = x.copy(), want.copy()
            bad_x[dp == 0] = np.uint64(0xFFFFFFFFFFFFFFFF)
            bad_p[pp == 0] = np.uint64(0xDEADBEEFDEADBEEF)
            d, q = to_dev(torch, bad_x), to_dev(torch, bad_p)
            enc.decode_prepare(dp, pp)
            enc.profile(True)
            enc.profile_reset()
            enc.decode(d, q)
            torch.cuda.synchronize()
            prof = enc.profile_read()
            enc.profile(False)
            if int((dp == 0).sum()) + int((pp == 0).sum()) > 32:  # (fewer: the direct path)
                assert prof.get("p61_tile_mid7_fold4", (0, 0, 0))[1] == 1, prof
                if logn in (11, 12):
                    assert any(name.endswith("_gather") for name in prof) and any(name.endswith("_scatter") for name in prof), prof
            assert (to_host(d).reshape(x.shape) == x).all(), (nlost, enc.plan())
            assert (to_host(q).reshape(want.shape) == bad_p).all()  # decode leaves the parity stripe alone
            enc.repair(d, q)
            torch.cuda.synchronize()
            assert (to_host(d).reshape(x.shape) == x).all() and (to_host(q).reshape(want.shape) == want).all(), nlost


@pytest.mark.parametrize("logn,elems", [(6, 3), (7, 70), (9, 64), (10, 66), (11, 2), (12, 20)])
def test_decode_of_the_n_equals_8k_code_is_folded(torch_cuda, fe, orc61, logn, elems):
    """n = 8k: of the decoder's 8k-point transform only the multiples of 8 are wanted; it runs as the big path's DIF passes, the folding MID tile
    (four consecutive positions into one: the second half of a size-2k path's 5-level MID) and that path's DIT passes — 2k outputs instead of 8k,
    the data at the even ones.  The profile shows the folding tile; round trip with the last tolerable loss (7k blocks) and with k blocks lost,
    parity against the oracle's composition."""
    torch = torch_cuda
    N, e = 1 << logn, 3
    rows = 7 * N
    x = rand_stripe(np.random.default_rng(8000 + logn), N, elems)
    want = p61_oracle_coset_parity(orc61, x, e)
    with fe.Encoder(N << e, N, 16 * elems, field=fe.FIELD_GF_P61_SQUARED) as enc:
        out = torch.empty(rows * 2 * elems, dtype=torch.int64, device="cuda:0")
        enc.encode(to_dev(torch, x), out)
        assert (to_host(out).reshape(rows, 2 * elems) == want).all()
        rng = np.random.default_rng(logn)
        for nlost in (7 * N, N):
            lost = rng.permutation(8 * N)[:nlost]
            dp, pp = np.ones(N, np.uint8), np.ones(rows, np.uint8)
            dp[lost[lost < N]] = 0
            pp[lost[lost >= N] - N] = 0
            if not (dp == 0).any():
                dp[0] = 0
                pp[np.flatnonzero(pp == 0)[0]] = 1
            bad_x, bad_p = x.copy(), want.copy()
            bad_x[dp == 0] = np.uint64(0xFFFFFFFFFFFFFFFF)
            bad_p[pp == 0] = np.uint64(0xDEADBEEFDEADBEEF)
            d, q = to_dev(torch, bad_x), to_dev(torch, bad_p)
            enc.set_option("decode_direct_max", 0)
            enc.decode_prepare(dp, pp)
            enc.profile(True)
            enc.profile_reset()
            enc.decode(d, q)
            torch.cuda.synchronize()
            prof = enc.profile_read()
            enc.profile(False)
            assert prof.get("p61_tile_mid7_fold4", (0, 0, 0))[1] == 1, prof
            assert (to_host(d).reshape(x.shape) == x).all(), (nlost, enc.plan())
            assert (to_host(q).reshape(want.shape) == bad_p).all()  # decode leaves the parity stripe alone
            enc.repair(d, q)
            torch.cuda.synchronize()
            assert (to_host(d).reshape(x.shape) == x).all() and (to_host(q).reshape(want.shape) == want).all(), nlost
        enc.set_option("decode_direct_max", 32)


@pytest.mark.parametrize("logn,e,elems", [(1, 2, 3), (4, 2, 70), (6, 3, 5), (10, 2, 9), (12, 3, 4), (14, 2, 2)])
def test_few_losses_in_n_equals_4k_and_8k_take_the_inner_codes_direct_path(torch_cuda, fe, orc61, logn, e, elems):
    """n = 4k / 8k with at most 32 blocks lost among the data and the FIRST coset: those 2k blocks are a (2k,k) code of their own, and its direct
    path rebuilds the data from 2k - few survivors — no transform over n runs (the profile is empty of the paths' kernels).  Lost parity blocks of
    the other cosets do not count against the 32; fastecc_repair re-encodes them.  Same bits as the transform path (decode_direct_max = 0) and the
    original stripes; one loss more and the transform path runs."""
    torch = torch_cuda
    N = 1 << logn
    rows = ((1 << e) - 1) * N
    rng = np.random.default_rng(700 + 10 * logn + e)
    x = rand_stripe(rng, N, elems)
    want = p61_oracle_coset_parity(orc61, x, e)
    with fe.Encoder(N << e, N, 16 * elems, field=fe.FIELD_GF_P61_SQUARED) as enc:
        cases = []  # (lost data, lost blocks of coset 0, lost blocks of the other cosets)
        for nd, n0, nother in ((1, 0, 0), (2, 3, 5), (9, 7, 40), (16, 0, rows), (10, 7, 0), (1, 16, 3), (20, 12, 9), (17, 16, 0)):
            nd, n0 = min(nd, N), min(n0, N)
            cases.append((nd, n0, max(0, min(nother, rows - N, rows - nd - n0))))  # (at least k blocks survive)
        for nd, n0, nother in cases:
            dp, pp = np.ones(N, np.uint8), np.ones(rows, np.uint8)
            dp[rng.permutation(N)[:nd]] = 0
            pp[rng.permutation(N)[:n0]] = 0
            pp[N + rng.permutation(rows - N)[:nother]] = 0
            bad_x, bad_p = x.copy(), want.copy()
            bad_x[dp == 0] = np.uint64(0xFFFFFFFFFFFFFFFF)
            bad_p[pp == 0] = np.uint64(0xDEADBEEFDEADBEEF)
            for direct_max in (32, 0):
                enc.set_option("decode_direct_max", direct_max)
                enc.decode_prepare(dp, pp)
                d, q = to_dev(torch, bad_x), to_dev(torch, bad_p)
                enc.profile(True)
                enc.profile_reset()
                enc.decode(d, q)
                torch.cuda.synchronize()
                prof = enc.profile_read()
                enc.profile(False)
                direct = direct_max == 32 and nd + n0 <= min(32, N)  # (k of the inner code's 2k blocks must survive)
                assert (len(prof) == 0) == direct, (nd, n0, nother, direct_max, prof)
                assert (to_host(d).reshape(x.shape) == x).all(), (nd, n0, nother, direct_max)
                assert (to_host(q).reshape(want.shape) == bad_p).all()  # decode leaves the parity stripe alone
                enc.repair(d, q)
                torch.cuda.synchronize()
                assert (to_host(d).reshape(x.shape) == x).all() and (to_host(q).reshape(want.shape) == want).all(), (nd, n0, nother, direct_max)
                hx, hp = bad_x.copy(), bad_p.copy()
                enc.repair(hx, hp, mem=fe.MEM_HOST)
                assert (hx == x).all() and (hp == want).all(), (nd, n0, nother, direct_max)
        enc.set_option("decode_direct_max", 32)


@pytest.mark.parametrize("logn", [1, 2, 5, 6, 7, 9, 12, 13, 14])
@pytest.mark.parametrize("e", [2, 3])
def test_multi_coset_parity_matches_oracle(torch_cuda, fe, orc61, logn, e):
    """More parity than data blocks over GF((2^61-1)^2): the n - k parity blocks are f on the 2^e - 1 cosets of the data points, k blocks per
    coset, codes nest (the first k blocks are the (2k,k) parity).  Sizes cover a MID-only plan, one and two tile chunks around MID."""
    N, elems = 1 << logn, (21 if logn < 12 else 66)
    x = rand_stripe(np.random.default_rng(61000 + 10 * logn + e), N, elems)
    want = p61_oracle_coset_parity(orc61, x, e)
    rows = ((1 << e) - 1) * N
    with fe.Encoder(N << e, N, 16 * elems, field=fe.FIELD_GF_P61_SQUARED) as enc:
        dx = to_dev(torch_cuda, x)
        out = torch_cuda.full((rows * 2 * elems,), 5, dtype=torch_cuda.int64, device="cuda:0")
        for _ in range(2):  # twice: the work stripe and the tables are reused
            enc.encode(dx, out)
            got = to_host(out).reshape(rows, 2 * elems)
            assert (got < P61).all()
            assert (got[:N] == orc61.encode(x)).all(), enc.plan()   # the codes nest
            assert (got == want).all(), enc.plan()
        assert (to_host(dx).reshape(x.shape) == x).all()
        host_out = np.empty_like(want)
        enc.encode(x, host_out, mem=fe.MEM_HOST)
        assert (host_out == want).all()
        with pytest.raises(fe.FastEccError):
            enc.encode(dx)  # in place is impossible: the parity is larger than the data
        # erasure decoding on the (k << e)-th roots of unity: any n - k blocks may go (data and parity), decode leaves the parity alone, repair
        # brings the lost parity blocks back too; erased blocks hold garbage
        rng = np.random.default_rng(logn * 100 + e)
        n = N << e
        for nlost in sorted({1, 3, min(n - N, 40), (n - N) // 2, n - N}):
            lost = rng.permutation(n)[:nlost]
            dp, pp = np.ones(N, np.uint8), np.ones(rows, np.uint8)
            dp[lost[lost < N]] = 0
            pp[lost[lost >= N] - N] = 0
            bad_x, bad_p = x.copy(), want.copy()
            bad_x[dp == 0] = np.uint64(0xFFFFFFFFFFFFFFFF)
            bad_p[pp == 0] = np.uint64(0xDEADBEEFDEADBEEF)
            enc.decode_prepare(dp, pp)
            d, q = to_dev(torch_cuda, bad_x), to_dev(torch_cuda, bad_p)
            enc.decode(d, q)
            torch_cuda.cuda.synchronize()
            assert (to_host(d).reshape(x.shape) == x).all(), (nlost, enc.plan())
            assert (to_host(q).reshape(want.shape) == bad_p).all(), nlost
            enc.repair(d, q)
            torch_cuda.cuda.synchronize()
            assert (to_host(d).reshape(x.shape) == x).all() and (to_host(q).reshape(want.shape) == want).all(), nlost
            if logn <= 9:
                hx, hp = bad_x.copy(), bad_p.copy()
                enc.repair(hx, hp, mem=fe.MEM_HOST)
                assert (hx == x).all() and (hp == want).all(), nlost
        with pytest.raises(fe.FastEccError) as ei:   # one block too many
            lost = rng.permutation(n)[: n - N + 1]
            dp, pp = np.ones(N, np.uint8), np.ones(rows, np.uint8)
            dp[lost[lost < N]] = 0
            pp[lost[lost >= N] - N] = 0
            enc.decode_prepare(dp, pp)
        assert ei.value.code == fe.E_INVAL


@pytest.mark.parametrize("plan", [2, 4, 13, 24])
def test_multi_coset_plans(torch_cuda, fe, orc61, plan):
    N, elems, e = 1 << 12, 40, 2
    x = rand_stripe(np.random.default_rng(61 + plan), N, elems)
    want = p61_oracle_coset_parity(orc61, x, e)
    with fe.Encoder(N << e, N, 16 * elems, field=fe.FIELD_GF_P61_SQUARED) as enc:
        enc.set_plan(plan)
        out = torch_cuda.empty(3 * N * 2 * elems, dtype=torch_cuda.int64, device="cuda:0")
        enc.encode(to_dev(torch_cuda, x), out)
        assert (to_host(out).reshape(3 * N, 2 * elems) == want).all(), enc.plan()


@pytest.mark.parametrize("logn,elems", [(11, 5), (12, 33), (13, 64), (14, 20), (16, 9)])
def test_decode_split_matches_the_folded_transform(torch_cuda, fe, orc61, logn, elems):
    """The even / odd split of the (2k,k) decoder (k >= 2^11: the data chain on a size-k path with the addend between the halves of MID, the parity
    half as a transform of k >> h rows) against the folded 2k-point transform (option "decode_split" = 0) and against the original data: patterns
    that make h = 5 .. 1, patterns whose parity losses sit ON the multiples of 2^h, data-only and parity-too losses, decode and repair; erased
    blocks hold garbage.  Bit-exact."""
    torch = torch_cuda
    N = 1 << logn
    rng = np.random.default_rng(4242 + logn)
    x = rand_stripe(rng, N, elems)
    dx = to_dev(torch, x)
    with encoder(fe, N, elems) as enc:
        par_dev = torch.empty_like(dx)
        enc.encode(dx, par_dev)
        par = to_host(par_dev).reshape(N, 2 * elems)
        if N * elems <= (1 << 17):
            assert (par == orc61.encode(x)).all()
        patterns = []
        for frac_d, frac_p in ((0.02, 0.0), (0.02, 0.02), (0.001, 0.001), (0.1, 0.0), (0.2, 0.0), (0.3, 0.0), (0.45, 0.0), (0.2, 0.2), (0.01, 0.6)):
            dp, pp = np.ones(N, np.uint8), np.ones(N, np.uint8)
            dp[rng.permutation(N)[: max(17, int(N * frac_d))]] = 0   # (more than 16 losses: not the direct path)
            pp[rng.permutation(N)[: int(N * frac_p)]] = 0
            patterns.append((dp, pp))
        dp, pp = np.ones(N, np.uint8), np.ones(N, np.uint8)          # every parity block at a multiple of 32 is gone: h must drop
        dp[rng.permutation(N)[:40]] = 0
        pp[::32] = 0
        patterns.append((dp, pp))
        dp, pp = np.ones(N, np.uint8), np.zeros(N, np.uint8)         # only the parity blocks at multiples of 32 survive, as many data blocks are lost
        pp[::32] = 1
        dp[rng.permutation(N)[: N // 32]] = 0
        patterns.append((dp, pp))
        for dp, pp in patterns:
            damaged, dpar = x.copy(), par.copy()
            damaged[dp == 0] = np.uint64(0xFFFFFFFFFFFFFFFF)
            dpar[pp == 0] = np.uint64(0xDEADBEEFDEADBEEF)
            got = {}
            for split in (1, 0):
                enc.set_option("decode_split", split)
                enc.decode_prepare(dp, pp)
                d, q = to_dev(torch, damaged), to_dev(torch, dpar)
                enc.decode(d, q)
                torch.cuda.synchronize()
                assert (to_host(d).reshape(x.shape) == x).all(), (split, int((dp == 0).sum()), int((pp == 0).sum()))
                assert (to_host(q).reshape(par.shape) == dpar).all()      # decode leaves the parity alone
                enc.repair(d, q)
                torch.cuda.synchronize()
                assert (to_host(d).reshape(x.shape) == x).all() and (to_host(q).reshape(par.shape) == par).all(), split
                hx, hp = damaged.copy(), dpar.copy()
                enc.repair(hx, hp, mem=fe.MEM_HOST)                       # host stripes through the same path
                assert (hx == x).all() and (hp == par).all(), split
            enc.set_option("decode_split", 1)


def test_decode_split_is_what_runs(torch_cuda, fe):
    """At k = 2^13 a 2 % loss runs the split: its kernels are the ones the profile shows (rows on the way in, the addend in MID, the scatter on the
    way out, the small transform of k / 32 rows), and none of the folded transform's."""
    torch = torch_cuda
    N, elems = 1 << 13, 64
    rng = np.random.default_rng(7)
    x = rand_stripe(rng, N, elems)
    dx = to_dev(torch, x)
    with encoder(fe, N, elems) as enc:
        par = torch.empty_like(dx)
        enc.encode(dx, par)
        dp, pp = np.ones(N, np.uint8), np.ones(N, np.uint8)
        dp[rng.permutation(N)[: N // 50]] = 0
        enc.decode_prepare(dp, pp)
        bad = dx.clone()
        bad.view(N, -1)[torch.from_numpy(np.flatnonzero(dp == 0)).to("cuda:0")] = -1
        enc.profile(True)
        enc.profile_reset()
        enc.decode(bad, par)
        torch.cuda.synchronize()
        names = set(enc.profile_read())
        enc.profile(False)
        assert torch.equal(bad, dx)
        assert any(n.endswith("_rows") for n in names) and any(n.endswith("_add") for n in names) and any(n.endswith("_scatter") for n in names), names
        assert not any(n.endswith("_gather") for n in names), names
        # fastecc_repair of a pattern that lost parity too: a second MID + DIT chain over the same two halves (MID's second half alone on the q~ the
        # data chain kept), no re-encode
        pp[rng.permutation(N)[: N // 50]] = 0
        enc.decode_prepare(dp, pp)
        bad, badp = dx.clone(), par.clone()
        bad.view(N, -1)[torch.from_numpy(np.flatnonzero(dp == 0)).to("cuda:0")] = -1
        badp.view(N, -1)[torch.from_numpy(np.flatnonzero(pp == 0)).to("cuda:0")] = -2
        enc.profile(True)
        enc.profile_reset()
        enc.repair(bad, badp)
        torch.cuda.synchronize()
        names = set(enc.profile_read())
        enc.profile(False)
        assert torch.equal(bad, dx) and torch.equal(badp, par)
        assert any(n.endswith("_up") for n in names) and not any(n.endswith("mid5") or n.endswith("mid6") or n.endswith("mid7") for n in names), names
