"""GPU tests (-m gpu) of the mixed-radix codes: transform order q * 2^m, q in {3, 5, 7, 9} (fastecc_create_ex with
FASTECC_CODE_MIXED_RADIX; NTT.md:43-46, the reference's NTT3/NTT9 codelets ntt.cpp:25-146).

Checker: the oracle's mixed-radix encode, itself pinned to the reference's Slow_NTT — the one reference transform that
takes such orders — and to its NTT3 / NTT9 codelets in tests/test_oracle.py.  Bit-exact, no tolerance."""
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


def rand_stripe(seed, N, S):
    x = np.random.default_rng(seed).integers(0, P, size=(N, S), dtype=np.uint64).astype(np.uint32)
    x.reshape(-1)[:4] = [0, 1, P - 1, 0x000FFFFF][: min(4, x.size)]
    return x


def test_order_selection(fe):
    assert [fe.mixed_radix_order(k) for k in (1, 2, 3, 5, 6, 7, 9, 11, 13, 17, 96, 97, 105, 1000, 393216, 393217, 430000)] == \
        [2, 2, 4, 6, 6, 8, 10, 12, 14, 18, 96, 104, 112, 1024, 393216, 425984, 458752]


@pytest.mark.parametrize("q", [3, 5, 7, 9, 13, 15])
@pytest.mark.parametrize("m,S", [(1, 1), (2, 7), (5, 64), (6, 33), (7, 256), (10, 128), (11, 40)])
def test_full_codes_match_the_oracle(torch_cuda, fe, oracle, q, m, S):
    """n = 2k, k = q * 2^m exactly: parity block j = f(w_2k^(2j+1)), data at the powers of w_k."""
    torch = torch_cuda
    k = q << m
    assert fe.mixed_radix_order(k) == k
    x = rand_stripe(q * 100 + m, k, S)
    want = oracle.encode_mixed(x)
    with fe.Encoder(2 * k, k, 4 * S, flags=fe.CODE_MIXED_RADIX) as enc:
        assert ("R%d:" % q) in enc.plan() or ("R%d+" % q) in enc.plan()
        d = to_dev(torch, x)
        out = torch.empty_like(d)
        enc.encode(d, out)
        torch.cuda.synchronize()
        assert np.array_equal(to_host(out).reshape(k, S), want), enc.plan()
        assert np.array_equal(to_host(d).reshape(k, S), x)  # out of place leaves the data alone
        enc.encode(d)  # in place, the reference's behaviour
        torch.cuda.synchronize()
        assert np.array_equal(to_host(d).reshape(k, S), want)


@pytest.mark.parametrize("k,m", [(3, 3), (5, 1), (11, 12), (100, 37), (111, 112), (1000, 24), (1500, 1536), (3000, 1)])
def test_any_k_zero_extension_and_fewer_parity_blocks(torch_cuda, fe, oracle, k, m):
    torch = torch_cuda
    S = 24
    order = fe.mixed_radix_order(k)
    x = rand_stripe(k + m, k, S)
    if order & (order - 1):
        want = oracle.encode_mixed_code(x, k + m, order)
    else:
        # a power of two is the ordinary fastecc_create code: the smallest power-of-two count >= m (at least order/16) of
        # evenly spaced parity blocks of the (2 order, order) code is computed, the first m are the parity
        count = max(order // 16, 1 << max(m - 1, 0).bit_length())
        z = np.zeros((order, S), dtype=np.uint32)
        z[:k] = x
        want = oracle.encode_fast(z)[:: order // count][:m]
    try:
        enc = fe.Encoder(k + m, k, 4 * S, flags=fe.CODE_MIXED_RADIX)
    except fe.FastEccError as e:
        assert m > order and e.code == fe.E_UNSUPPORTED
        return
    with enc:
        d = to_dev(torch, x)
        out = torch.full((m * S,), 0x77777777, dtype=torch.int32, device="cuda:0")
        enc.encode(d, out)
        torch.cuda.synchronize()
        assert np.array_equal(to_host(out).reshape(m, S), want), enc.plan()
        host_out = np.empty((m, S), dtype=np.uint32)
        enc.encode_host(x, host_out)
        assert np.array_equal(host_out, want)


def test_every_plan_and_block_pointers(torch_cuda, fe, oracle):
    torch = torch_cuda
    k, S = 3 << 11, 96
    x = rand_stripe(5, k, S)
    want = oracle.encode_mixed(x)
    with fe.Encoder(2 * k, k, 4 * S, flags=fe.CODE_MIXED_RADIX) as enc:
        for plan in (0, 51, 42, 34, 1100, 2080, 3100):
            enc.set_plan(plan)
            d = to_dev(torch, x)
            enc.encode(d)
            torch.cuda.synchronize()
            assert np.array_equal(to_host(d).reshape(k, S), want), (plan, enc.plan())
        blocks = [np.ascontiguousarray(x[i]).copy() for i in range(k)]
        enc.encode_blocks([b.ctypes.data for b in blocks])
        assert np.array_equal(np.stack(blocks), want)
        assert enc.check_range(to_dev(torch, x)) == 0
        for call in (lambda: enc.ntt(to_dev(torch, x)), lambda: enc.encode_columns(d, d, 0, 32)):
            with pytest.raises(fe.FastEccError) as ei:
                call()
            assert ei.value.code == fe.E_UNSUPPORTED


def test_flags_validation(torch_cuda, fe):
    with pytest.raises(fe.FastEccError) as ei:
        fe.Encoder(8, 4, 64, flags=8)
    assert ei.value.code == fe.E_INVAL
    with pytest.raises(fe.FastEccError) as ei:
        fe.Encoder(2 * 96, 96, 64, field=fe.FIELD_GF_P61_SQUARED, flags=fe.CODE_MIXED_RADIX)
    assert ei.value.code == fe.E_UNSUPPORTED
    with fe.Encoder(256, 128, 64, flags=fe.CODE_MIXED_RADIX) as enc:  # a power of two: the ordinary context
        assert "R" not in enc.plan().split(" v")[0]


def test_large_orders_linearity_and_sampled_columns(torch_cuda, fe, oracle):
    """k = 3 * 2^17 and 9 * 2^16 x 4 KB (1.5 and 2.25 GiB stripes): eight columns re-encoded by the oracle, and linearity."""
    torch = torch_cuda
    S = 1024
    for q, m in ((3, 17), (9, 16)):
        k = q << m
        g = torch.Generator(device="cuda:0")
        g.manual_seed(q)
        a = torch.randint(0, P, (k * S,), dtype=torch.int64, device="cuda:0", generator=g).to(torch.int32)
        b = torch.randint(0, P, (k * S,), dtype=torch.int64, device="cuda:0", generator=g).to(torch.int32)
        pa, pb, ps = torch.empty_like(a), torch.empty_like(a), torch.empty_like(a)
        with fe.Encoder(2 * k, k, 4 * S, flags=fe.CODE_MIXED_RADIX) as enc:
            enc.encode(a, pa)
            enc.encode(b, pb)
            ua = a.to(torch.int64) & 0xFFFFFFFF
            ub = b.to(torch.int64) & 0xFFFFFFFF
            s = ((ua + ub) % P).to(torch.int32)
            del ua, ub
            enc.encode(s, ps)
            torch.cuda.synchronize()
        lhs = ((pa.to(torch.int64) & 0xFFFFFFFF) + (pb.to(torch.int64) & 0xFFFFFFFF)) % P
        assert torch.equal(lhs, ps.to(torch.int64) & 0xFFFFFFFF)
        del lhs, s, ps, pb, b
        cols = [0, 1, 31, 32, 500, 777, 1022, 1023]
        sample = to_host(a.view(k, S)[:, cols].contiguous()).reshape(k, len(cols))
        want = oracle.encode_mixed(sample)
        assert np.array_equal(to_host(pa.view(k, S)[:, cols].contiguous()).reshape(k, len(cols)), want), (q, m)
        del a, pa
        torch.cuda.empty_cache()


def erase_and_check(torch, fe, enc, x, par, lost_data, lost_parity, repair):
    """Overwrite the lost blocks with garbage, decode (or repair) on the device, compare with the originals."""
    k, S = x.shape
    m = par.shape[0]
    dp = np.ones(k, dtype=np.uint8)
    pp = np.ones(m, dtype=np.uint8)
    dp[lost_data] = 0
    pp[lost_parity] = 0
    bad_x, bad_p = x.copy(), par.copy()
    bad_x[lost_data] = 0xEEEEEEEE
    bad_p[lost_parity] = 0xDDDDDDDD
    d, q = to_dev(torch, bad_x), to_dev(torch, bad_p)
    enc.decode_prepare(dp, pp)
    if repair:
        enc.repair(d, q)
    else:
        enc.decode(d, q)
    torch.cuda.synchronize()
    assert np.array_equal(to_host(d).reshape(k, S), x)
    if repair:
        assert np.array_equal(to_host(q).reshape(m, S), par)
    else:
        assert np.array_equal(to_host(q).reshape(m, S), bad_p)  # decode leaves the parity stripe alone


@pytest.mark.parametrize("q", [3, 5, 7, 9, 13, 15])
@pytest.mark.parametrize("m,S", [(1, 3), (3, 64), (6, 40), (10, 32)])
def test_decoder_of_mixed_radix_codes(torch_cuda, fe, oracle, q, m, S):
    """Round trip through the encoder that test_full_codes_match_the_oracle pins: encode, lose blocks, decode = original.
    Patterns: one block, every data block (the most the code tolerates), random halves of data and parity."""
    torch = torch_cuda
    k = q << m
    rng = np.random.default_rng(q * 1000 + m)
    x = rand_stripe(q * 10 + m, k, S)
    with fe.Encoder(2 * k, k, 4 * S, flags=fe.CODE_MIXED_RADIX) as enc:
        out = torch.empty(k * S, dtype=torch.int32, device="cuda:0")
        enc.encode(to_dev(torch, x), out)
        torch.cuda.synchronize()
        par = to_host(out).reshape(k, S).copy()
        if k <= 96:
            assert np.array_equal(par, oracle.encode_mixed(x))
        none = np.array([], dtype=np.int64)
        erase_and_check(torch, fe, enc, x, par, np.array([k - 1]), none, False)
        erase_and_check(torch, fe, enc, x, par, np.arange(k), none, False)
        for repair in (False, True):
            lost = rng.permutation(2 * k)[:k]  # exactly n - k losses over the whole codeword
            erase_and_check(torch, fe, enc, x, par, lost[lost < k], lost[lost >= k] - k, repair)
        few = rng.permutation(2 * k)[: max(1, k // 7)]
        erase_and_check(torch, fe, enc, x, par, few[few < k], few[few >= k] - k, True)
        # one loss too many is refused
        dp = np.zeros(k, dtype=np.uint8)
        pp = np.ones(k, dtype=np.uint8)
        pp[0] = 0
        with pytest.raises(fe.FastEccError) as ei:
            enc.decode_prepare(dp, pp)
        assert ei.value.code == fe.E_INVAL


@pytest.mark.parametrize("k,m", [(5, 3), (11, 12), (100, 37), (1000, 900), (1500, 1536), (3000, 100)])
def test_decoder_of_zero_extended_mixed_codes(torch_cuda, fe, k, m):
    """Any k with a mixed-radix order: data blocks k..order-1 are known zeros, parity blocks m..order-1 do not exist (they count
    as lost), so exactly m losses among the k + m blocks are repaired."""
    torch = torch_cuda
    S = 24
    order = fe.mixed_radix_order(k)
    if not order & (order - 1):
        pytest.skip("a power of two: the ordinary code, tests/test_gpu_decode.py")
    rng = np.random.default_rng(k * 7 + m)
    x = rand_stripe(k + m, k, S)
    with fe.Encoder(k + m, k, 4 * S, flags=fe.CODE_MIXED_RADIX) as enc:
        out = torch.empty(m * S, dtype=torch.int32, device="cuda:0")
        enc.encode(to_dev(torch, x), out)
        torch.cuda.synchronize()
        par = to_host(out).reshape(m, S).copy()
        for repair in (False, True):
            lost = rng.permutation(k + m)[:m]
            erase_and_check(torch, fe, enc, x, par, lost[lost < k], lost[lost >= k] - k, repair)
        erase_and_check(torch, fe, enc, x, par, np.arange(min(k, m)), np.array([], dtype=np.int64), False)
        # host stripes
        dp = np.ones(k, dtype=np.uint8)
        pp = np.ones(m, dtype=np.uint8)
        dp[0] = 0
        pp[m - 1] = 0
        hx, hp = x.copy(), par.copy()
        hx[0] = 1
        hp[m - 1] = 2
        enc.decode_prepare(dp, pp)
        enc.repair(hx, hp, mem=fe.MEM_HOST)
        assert np.array_equal(hx, x) and np.array_equal(hp, par)


def test_decoder_at_a_large_mixed_order(torch_cuda, fe):
    """k = 3 * 2^17 x 4 KB blocks (1.5 GiB of data): 30 % of the codeword lost, repaired, compared on the device."""
    torch = torch_cuda
    S, q, m = 1024, 3, 17
    k = q << m
    g = torch.Generator(device="cuda:0")
    g.manual_seed(99)
    x = torch.randint(0, P, (k * S,), generator=g, device="cuda:0", dtype=torch.int64).to(torch.int32)
    with fe.Encoder(2 * k, k, 4 * S, flags=fe.CODE_MIXED_RADIX) as enc:
        par = torch.empty_like(x)
        enc.encode(x, par)
        rng = np.random.default_rng(3)
        lost = rng.permutation(2 * k)[: int(0.3 * 2 * k)]
        ld, lp = lost[lost < k], lost[lost >= k] - k
        dp = np.ones(k, dtype=np.uint8)
        pp = np.ones(k, dtype=np.uint8)
        dp[ld] = 0
        pp[lp] = 0
        bx, bp = x.clone(), par.clone()
        bx.view(k, S)[torch.from_numpy(ld).to("cuda:0")] = -1
        bp.view(k, S)[torch.from_numpy(lp).to("cuda:0")] = -2
        enc.decode_prepare(dp, pp)
        enc.repair(bx, bp)
        torch.cuda.synchronize()
        assert torch.equal(bx, x) and torch.equal(bp, par)


@pytest.mark.parametrize("q", [3, 5, 7, 9, 13, 15])
@pytest.mark.parametrize("m", [11, 12, 13, 14, 15, 16, 17, 18])
def test_fused_odd_radix_level(torch_cuda, fe, oracle, q, m):
    """Orders with an outer tile above MID: the odd-radix level fused into that tile (3 trips) against its own two passes
    (5 trips, option fuse_radix = 0) and, where the oracle finishes quickly, against the oracle; full and zero-extended codes."""
    torch = torch_cuda
    order = q << m
    S = 4 if m >= 16 else 20
    for k, par in ((order, order), (order - 7, order // 2 + 3)):
        x = rand_stripe(q * 100 + m, k, S)
        with fe.Encoder(k + par, k, 4 * S, flags=fe.CODE_MIXED_RADIX) as enc:
            fused_plan = enc.plan()
            d = to_dev(torch, x)
            out = torch.empty(par * S, dtype=torch.int32, device="cuda:0")
            enc.encode(d, out)
            torch.cuda.synchronize()
            got = to_host(out).reshape(par, S).copy()
            enc.set_option("fuse_radix", 0)
            assert ("R%d:dif1" % q) in enc.plan()
            out.zero_()
            enc.encode(d, out)
            torch.cuda.synchronize()
            assert np.array_equal(to_host(out).reshape(par, S), got), (fused_plan, enc.plan())
            enc.set_option("fuse_radix", 1)
            assert enc.plan() == fused_plan
            if k == order:
                work = to_dev(torch, x)
                enc.encode(work)  # in place
                torch.cuda.synchronize()
                assert np.array_equal(to_host(work).reshape(k, S), got), fused_plan
            if m <= 13:
                assert np.array_equal(got, oracle.encode_mixed_code(x, k + par, order)), fused_plan
        supported = m - 10 <= {3: 8, 5: 7, 7: 7, 9: 6, 13: 6, 15: 6}[q]
        assert (("R%d+dif" % q) in fused_plan) == supported, fused_plan


@pytest.mark.parametrize("m", [12, 13, 15, 17])
def test_power_of_two_top_level_as_a_radix(torch_cuda, fe, oracle, m):
    """FASTECC_CODE_TOP_RADIX2 (an A/B experiment): the same (2k,k) code, its top level through the odd-radix machinery with q = 2,
    fused and unfused; bit-identical to fastecc_create's path and to the oracle."""
    torch = torch_cuda
    k, S = 1 << m, 12
    x = rand_stripe(m, k, S)
    d = to_dev(torch, x)
    with fe.Encoder(2 * k, k, 4 * S) as ref:
        want = torch.empty_like(d)
        ref.encode(d, want)
    if m <= 13:
        assert np.array_equal(to_host(want).reshape(k, S), oracle.encode_fast(x))
    with fe.Encoder(2 * k, k, 4 * S, flags=fe.CODE_TOP_RADIX2) as enc:
        assert "R2+dif" in enc.plan()
        out = torch.empty_like(d)
        enc.encode(d, out)
        torch.cuda.synchronize()
        assert torch.equal(out, want), enc.plan()
        enc.set_option("fuse_radix", 0)
        assert "R2:dif1" in enc.plan()
        out.zero_()
        enc.encode(d, out)
        torch.cuda.synchronize()
        assert torch.equal(out, want), enc.plan()
    for bad in ((2 * k + 2, k + 1), (3 * k, k), (64, 32)):
        with pytest.raises(fe.FastEccError) as ei:
            fe.Encoder(bad[0], bad[1], 4 * S, flags=fe.CODE_TOP_RADIX2)
        assert ei.value.code == fe.E_UNSUPPORTED


@pytest.mark.parametrize("k,m", [(96, 96), (100, 37), (3 << 11, 3 << 11), (5000, 1200), (7 << 9, 100)])
def test_few_losses_of_mixed_radix_codes(torch_cuda, fe, k, m):
    """Up to 16 lost blocks: interpolated from the surviving data blocks and a few parity blocks (the data points are the (q 2^m)-th roots of
    unity, the formulas do not care about the radix); both decoder paths agree."""
    torch = torch_cuda
    S = 12
    rng = np.random.default_rng(k + m)
    x = rand_stripe(k + m, k, S)
    with fe.Encoder(k + m, k, 4 * S, flags=fe.CODE_MIXED_RADIX) as enc:
        out = torch.empty(m * S, dtype=torch.int32, device="cuda:0")
        enc.encode(to_dev(torch, x), out)
        torch.cuda.synchronize()
        par = to_host(out).reshape(m, S).copy()
        none = np.array([], dtype=np.int64)
        for e in (1, 3, 16, 17):
            if e > m:
                continue
            lost = np.unique(np.r_[int(rng.integers(0, k)), rng.permutation(k + m)[: e - 1]])
            for direct_max in (16, 0):
                enc.set_option("decode_direct_max", direct_max)
                erase_and_check(torch, fe, enc, x, par, lost[lost < k], lost[lost >= k] - k, True)
                erase_and_check(torch, fe, enc, x, par, lost[lost < k], none, False)
        enc.set_option("decode_direct_max", 16)


@pytest.mark.parametrize("k", [5, 96, 1000, 3 << 11, 9 << 10])
def test_few_parity_blocks_of_mixed_radix_codes(torch_cuda, fe, oracle, k):
    """n - k <= 8 on a mixed-radix order: the parity straight from the Lagrange basis of the (q 2^m)-th roots of unity; same bits as the
    pipeline (encode_direct_max = 0) and the oracle."""
    torch = torch_cuda
    S = 12
    order = fe.mixed_radix_order(k)
    if not order & (order - 1):
        pytest.skip("a power of two: tests/test_gpu_general.py")
    x = rand_stripe(k, k, S)
    for m in (1, 3, 8):
        if m > order:
            continue
        want = oracle.encode_mixed_code(x, k + m, order)
        with fe.Encoder(k + m, k, 4 * S, flags=fe.CODE_MIXED_RADIX) as enc:
            for direct_max in (8, 0):
                enc.set_option("encode_direct_max", direct_max)
                out = torch.zeros(m * S, dtype=torch.int32, device="cuda:0")
                enc.encode(to_dev(torch, x), out)
                torch.cuda.synchronize()
                assert np.array_equal(to_host(out).reshape(m, S), want), (k, m, direct_max, enc.plan())


def test_orders_above_2_20(torch_cuda, fe):
    """k = 3 * 2^19 (transform order 1.5 M > 2^20; the code tolerates 1.5 M losses): up to 256 lost blocks take the direct path, more go
    through the locator tree padded to 2^20 roots (the largest cyclic product this field has), and patterns beyond 2^20 erasures are refused."""
    torch = torch_cuda
    k, S = 3 << 19, 2
    g = torch.Generator(device="cuda:0")
    g.manual_seed(5)
    x = torch.randint(0, P, (k * S,), generator=g, device="cuda:0", dtype=torch.int64).to(torch.int32)
    with fe.Encoder(2 * k, k, 4 * S, flags=fe.CODE_MIXED_RADIX) as enc:
        par = torch.empty_like(x)
        enc.encode(x, par)
        rng = np.random.default_rng(8)
        for count in (1, 16, 17, 200, 257, 5000, (1 << 20) + 1):
            lost = rng.permutation(2 * k)[:count]
            lost[0] = int(rng.integers(0, k))
            dp, pp = np.ones(k, np.uint8), np.ones(k, np.uint8)
            dp[lost[lost < k]] = 0
            pp[lost[lost >= k] - k] = 0
            if count > (1 << 20):
                with pytest.raises(fe.FastEccError) as ei:
                    enc.decode_prepare(dp, pp)
                assert ei.value.code == fe.E_UNSUPPORTED
                continue
            bx, bp = x.clone(), par.clone()
            bx.view(k, S)[torch.from_numpy(np.flatnonzero(dp == 0)).to("cuda:0")] = -1
            bp.view(k, S)[torch.from_numpy(np.flatnonzero(pp == 0)).to("cuda:0")] = -2
            enc.decode_prepare(dp, pp)
            enc.repair(bx, bp)
            torch.cuda.synchronize()
            assert torch.equal(bx, x) and torch.equal(bp, par), count
