"""GPU tests (-m gpu) of the mixed-radix codes whose odd factor is a product of coprime factors: transform order q * 2^m,
q in {21, 35, 39, 45, 63, 65, 91, 105, 117} (fastecc_create_ex with FASTECC_CODE_MIXED_RADIX_PFA; NTT.md:43-46 "PFA NTT as
well as NTT kernels of orders 3,5,7,9,13").

Checker: the oracle's mixed-radix encode (any odd q by the definition), pinned to the reference's Slow_NTT at these orders
in tests/test_oracle.py.  Bit-exact, no tolerance."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

P = 0xFFF00001
PFA_Q = [21, 35, 39, 45, 63, 65, 91, 105, 117]


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
    ks = (1, 40, 41, 43, 67, 71, 79, 100, 129, 131, 1100, 2700, 3400, 40000, 600000)
    assert [fe.mixed_radix_order(k, pfa=True) for k in ks] == [2, 40, 42, 48, 70, 72, 80, 104, 130, 140, 1120, 2880, 3584, 40960, 638976]
    for k in ks:
        assert fe.mixed_radix_order(k, pfa=True) <= fe.mixed_radix_order(k)
    # the next order is never more than 8.4 % away (20 % with the seven plain factors) once 2^m >= 64
    worst = max(fe.mixed_radix_order(k, pfa=True) / k for k in range(4096, 8193))
    assert worst < 1.0834, worst
    assert max(fe.mixed_radix_order(k) / k for k in range(4096, 8193)) > 1.19


@pytest.mark.parametrize("q", PFA_Q)
@pytest.mark.parametrize("m,S", [(1, 1), (2, 7), (5, 64), (6, 33), (10, 20)])
def test_full_codes_match_the_oracle(torch_cuda, fe, oracle, q, m, S):
    """n = 2k, k = q * 2^m exactly: parity block j = f(w_2k^(2j+1)), data at the powers of w_k; up to 2^10 the odd-radix level is
    the pass around the MID tile (three trips)."""
    torch = torch_cuda
    k = q << m
    assert fe.mixed_radix_order(k, pfa=True) == k
    x = rand_stripe(q * 100 + m, k, S)
    want = oracle.encode_mixed(x)
    with fe.Encoder(2 * k, k, 4 * S, flags=fe.CODE_MIXED_RADIX_PFA) as enc:
        assert ("R%d:" % q) in enc.plan() or ("R%d+" % q) in enc.plan()
        d = to_dev(torch, x)
        out = torch.empty_like(d)
        enc.encode(d, out)
        torch.cuda.synchronize()
        assert np.array_equal(to_host(out).reshape(k, S), want), enc.plan()
        assert np.array_equal(to_host(d).reshape(k, S), x)
        enc.encode(d)  # in place
        torch.cuda.synchronize()
        assert np.array_equal(to_host(d).reshape(k, S), want)


@pytest.mark.parametrize("q,m,S", [(21, 11, 24), (21, 13, 65), (21, 16, 64), (35, 12, 16), (35, 15, 8), (39, 11, 70), (39, 14, 8), (45, 13, 12), (45, 15, 4),
                                   (63, 11, 8), (63, 14, 5), (63, 15, 2), (65, 11, 6), (117, 12, 2)])
def test_orders_above_2_10_fused_and_not(torch_cuda, fe, oracle, q, m, S):
    """Above 2^10 blocks per stripe the odd-radix level rides on the outer tile pass where a fused shape exists (q <= 63, up to 3-6
    outer levels) and has its own two passes otherwise; option fuse_radix = 0 forces the latter: same parity either way."""
    torch = torch_cuda
    k = q << m
    x = rand_stripe(q * 1000 + m, k, S)
    want = oracle.encode_mixed(x)
    with fe.Encoder(2 * k, k, 4 * S, flags=fe.CODE_MIXED_RADIX_PFA) as enc:
        fused = ("R%d+" % q) in enc.plan()
        most = 0 if q > 63 else 6 if q == 21 else 3 if q == 63 else 4
        assert fused == (m - 10 <= most), enc.plan()
        d = to_dev(torch, x)
        out = torch.empty_like(d)
        enc.encode(d, out)
        torch.cuda.synchronize()
        assert np.array_equal(to_host(out).reshape(k, S), want), enc.plan()
        if fused:
            enc.set_option("fuse_radix", 0)
            assert ("R%d:" % q) in enc.plan()
            out.zero_()
            enc.encode(d, out)
            torch.cuda.synchronize()
            assert np.array_equal(to_host(out).reshape(k, S), want), enc.plan()


@pytest.mark.parametrize("k,m", [(41, 12), (67, 70), (131, 40), (1100, 300), (2700, 2880), (3400, 17), (40000, 100)])
def test_any_k_zero_extension_and_fewer_parity_blocks(torch_cuda, fe, oracle, k, m):
    torch = torch_cuda
    S = 12
    order = fe.mixed_radix_order(k, pfa=True)
    assert order & (order - 1)
    x = rand_stripe(k + m, k, S)
    want = oracle.encode_mixed_code(x, k + m, order)
    with fe.Encoder(k + m, k, 4 * S, flags=fe.CODE_MIXED_RADIX_PFA) as enc:
        d = to_dev(torch, x)
        out = torch.full((m * S,), 0x77777777, dtype=torch.int32, device="cuda:0")
        enc.encode(d, out)
        torch.cuda.synchronize()
        assert np.array_equal(to_host(out).reshape(m, S), want), enc.plan()
        host_out = np.empty((m, S), dtype=np.uint32)
        enc.encode_host(x, host_out)
        assert np.array_equal(host_out, want)


@pytest.mark.parametrize("q,m,S", [(21, 4, 9), (35, 7, 16), (39, 3, 5), (45, 11, 8), (63, 6, 3), (65, 5, 4), (91, 2, 7), (105, 8, 2), (117, 4, 6)])
def test_decoder_of_composite_orders(torch_cuda, fe, oracle, q, m, S):
    """Erasure decoding at these orders: a quarter of the codeword lost (the transform decoder with its 2 N1-point transforms) and
    three lost blocks (the direct path); data and parity restored bit for bit."""
    torch = torch_cuda
    k = q << m
    x = rand_stripe(q + m, k, S)
    with fe.Encoder(2 * k, k, 4 * S, flags=fe.CODE_MIXED_RADIX_PFA) as enc:
        d = to_dev(torch, x)
        par = torch.empty_like(d)
        enc.encode(d, par)
        torch.cuda.synchronize()
        want_p = to_host(par).reshape(k, S).copy()
        assert np.array_equal(want_p, oracle.encode_mixed(x))
        rng = np.random.default_rng(q * m)
        for lost in (k // 2, 3):
            idx = rng.choice(2 * k, size=lost, replace=False)
            dp = np.ones(k, np.uint8)
            pp = np.ones(k, np.uint8)
            dp[idx[idx < k]] = 0
            pp[idx[idx >= k] - k] = 0
            bx = x.copy()
            bx[dp == 0] = 0x5A5A5A5A % P
            bp = want_p.copy()
            bp[pp == 0] = 0x3C3C3C3C % P
            dx, dq = to_dev(torch, bx), to_dev(torch, bp)
            enc.decode_prepare(dp, pp)
            enc.repair(dx, dq)
            torch.cuda.synchronize()
            assert np.array_equal(to_host(dx).reshape(k, S), x), (lost, enc.plan())
            assert np.array_equal(to_host(dq).reshape(k, S), want_p), (lost, enc.plan())


@pytest.mark.parametrize("seed", range(40))
def test_random_configuration(torch_cuda, fe, oracle, seed):
    """Seeded random (n,k), block size and loss pattern with the composite orders on: encode against the oracle, damage, repair."""
    torch = torch_cuda
    rng = np.random.default_rng(7000 + seed)
    k = int(rng.integers(3, 9000))
    order = fe.mixed_radix_order(k, pfa=True)
    m = int(rng.integers(1, order + 1))
    S = int(rng.choice([1, 2, 3, 5, 8, 17, 32, 33, 64]))
    if k * S > 300000:
        S = max(1, 300000 // k)
    x = rng.integers(0, P, size=(k, S), dtype=np.uint64).astype(np.uint32)
    if order & (order - 1):
        want = oracle.encode_mixed_code(x, k + m, order)
    else:
        count = max(order // 16, 1 << max(m - 1, 0).bit_length())
        z = np.zeros((order, S), dtype=np.uint32)
        z[:k] = x
        want = oracle.encode_fast(z)[:: order // count][:m]
    with fe.Encoder(k + m, k, 4 * S, flags=fe.CODE_MIXED_RADIX_PFA) as enc:
        what = (k, m, S, order, enc.plan())
        d = to_dev(torch, x)
        out = torch.full((m * S,), 0x55555555, dtype=torch.int32, device="cuda:0")
        enc.encode(d, out)
        torch.cuda.synchronize()
        assert np.array_equal(to_host(out).reshape(m, S), want), what
        lost = int(rng.integers(1, m + 1))
        idx = rng.choice(k + m, size=lost, replace=False)
        dp = np.ones(k, np.uint8)
        pp = np.ones(m, np.uint8)
        dp[idx[idx < k]] = 0
        pp[idx[idx >= k] - k] = 0
        bx, bp = x.copy(), want.copy()
        bx[dp == 0] = 0x11111111
        bp[pp == 0] = 0x22222222
        dx, dq = to_dev(torch, bx), to_dev(torch, bp)
        enc.decode_prepare(dp, pp)
        enc.repair(dx, dq)
        torch.cuda.synchronize()
        assert np.array_equal(to_host(dx).reshape(k, S), x), what + (lost,)
        assert np.array_equal(to_host(dq).reshape(m, S), want), what + (lost,)
