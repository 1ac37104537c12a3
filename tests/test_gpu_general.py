"""GPU parity tests (-m gpu) for arbitrary (n,k) by zero extension (RS.md:23-33, steps 1-7 of the reference's own
description): k data blocks are the first k of N = 2^ceil(log2 k), the parity is the first n-k blocks of the
(N + M, N) code.  The checker is the PINNED oracle on the zero-padded stripe, subsampled — i.e. a subset of what
the unmodified reference computes for the padded input.  Bit-exact."""
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


def to_host(t, shape):
    return t.cpu().numpy().view(np.uint32).reshape(shape)


def expected(oracle, x, m):
    k, S = x.shape
    lg = max(1, int(np.ceil(np.log2(k))))
    N = 1 << lg
    lgm = int(np.ceil(np.log2(m))) if m > 1 else 0
    fold = min(lg - lgm, 4)
    padded = np.zeros((N, S), dtype=np.uint32)
    padded[:k] = x
    return oracle.encode_fast(padded)[:: 1 << fold][:m]


@pytest.mark.parametrize("k,m", [(1, 1), (3, 2), (5, 5), (10, 4), (100, 30), (100, 128), (1000, 300), (1024, 1000), (1025, 1),
                                 (3000, 3000), (5000, 1200), (10000, 2000), (65, 64)])
def test_any_n_k(torch_cuda, fe, oracle, k, m):
    S = 37 if k < 4000 else 16
    x = np.random.default_rng(k * 7 + m).integers(0, P, size=(k, S), dtype=np.uint64).astype(np.uint32)
    want = expected(oracle, x, m)
    with fe.Encoder(k + m, k, 4 * S) as enc:
        dx = to_dev(torch_cuda, x)
        out = torch_cuda.empty(m * S, dtype=torch_cuda.int32, device="cuda:0")
        enc.encode(dx, out)
        assert (to_host(out, (m, S)) == want).all(), enc.plan()
        assert (to_host(dx, (k, S)) == x).all()
        assert enc.check_range(dx) == 0
        host_out = np.empty((m, S), dtype=np.uint32)
        enc.encode_host(x, host_out)
        assert (host_out == want).all()
        if m <= k:
            enc.encode(dx)  # in place: the first n - k data blocks are replaced by the parity
            got = to_host(dx, (k, S))
            assert (got[:m] == want).all() and (got[m:] == x[m:]).all()
            blocks = [np.ascontiguousarray(x[i]).copy() for i in range(k)]
            enc.encode_blocks([b.ctypes.data for b in blocks])
            assert (np.stack(blocks[:m]) == want).all()
        else:
            with pytest.raises(fe.FastEccError):
                enc.encode(dx)
        if k & (k - 1):   # the stand-alone transform needs a power-of-two length
            with pytest.raises(fe.FastEccError) as ei:
                enc.ntt(dx)
            assert ei.value.code == fe.E_UNSUPPORTED
        # lose as many blocks as there are parity blocks, all over the codeword, and get the data back
        rng = np.random.default_rng(k + m)
        lost = rng.permutation(k + m)[:m]
        dp, pp = np.ones(k, np.uint8), np.ones(m, np.uint8)
        dp[lost[lost < k]] = 0
        pp[lost[lost >= k] - k] = 0
        damaged = x.copy()
        damaged[dp == 0] = 0xFFFFFFFF
        dpar = want.copy()
        dpar[pp == 0] = 0x12345678
        enc.decode_prepare(dp, pp)
        dd = to_dev(torch_cuda, damaged)
        enc.decode(dd, to_dev(torch_cuda, dpar))
        assert (to_host(dd, (k, S)) == x).all()
        host = damaged.copy()
        enc.decode(host, dpar, mem=fe.MEM_HOST)
        assert (host == x).all()
        if m < k + m - 1 and m + 1 <= k:
            dp2 = dp.copy()
            dp2[np.flatnonzero(dp2)[0]] = 0          # one erasure too many
            with pytest.raises(fe.FastEccError) as ei:
                enc.decode_prepare(dp2, pp)
            assert ei.value.code == fe.E_INVAL


def test_matches_the_unmodified_reference_on_the_padded_stripe(torch_cuda, fe):
    from oracle import Reference
    if not Reference.available():
        pytest.skip("oracle/_ref not built")
    ref = Reference()
    k, m, S = 3000, 1000, 64
    x = np.random.default_rng(2).integers(0, P, size=(k, S), dtype=np.uint64).astype(np.uint32)
    padded = np.zeros((4096, S), dtype=np.uint32)
    padded[:k] = x
    want = ref.encode(padded)[::4][:m]
    with fe.Encoder(k + m, k, 4 * S) as enc:
        out = torch_cuda.empty(m * S, dtype=torch_cuda.int32, device="cuda:0")
        enc.encode(to_dev(torch_cuda, x), out)
        assert (to_host(out, (m, S)) == want).all()


def test_random_codes_round_trip(torch_cuda, fe, oracle):
    """Forty random (n, k, block size, erasure pattern) combinations: encode == oracle on the padded stripe, then
    lose up to n - k blocks and decode back."""
    torch = torch_cuda
    rng = np.random.default_rng(20240926)
    for case in range(40):
        k = int(rng.integers(1, 3000))
        N = 1 << max(1, int(np.ceil(np.log2(k))))
        m = int(rng.integers(1, N + 1))
        S = int(rng.choice([1, 2, 5, 16, 33, 64, 100]))
        x = rng.integers(0, P, size=(k, S), dtype=np.uint64).astype(np.uint32)
        want = expected(oracle, x, m)
        with fe.Encoder(k + m, k, 4 * S) as enc:
            out = torch.empty(m * S, dtype=torch.int32, device="cuda:0")
            enc.encode(to_dev(torch, x), out)
            assert (to_host(out, (m, S)) == want).all(), (case, k, m, S, enc.plan())
            nlost = int(rng.integers(0, m + 1))
            lost = rng.permutation(k + m)[:nlost]
            dp, pp = np.ones(k, np.uint8), np.ones(m, np.uint8)
            dp[lost[lost < k]] = 0
            pp[lost[lost >= k] - k] = 0
            damaged = x.copy()
            damaged[dp == 0] = 0xA5A5A5A5
            enc.decode_prepare(dp, pp)
            dd = to_dev(torch, damaged)
            enc.decode(dd, out)
            assert (to_host(dd, (k, S)) == x).all(), (case, k, m, S, nlost)


@pytest.mark.parametrize("k", [1, 2, 7, 64, 1000, 4096, 100000])
def test_few_parity_blocks_are_encoded_directly(torch_cuda, fe, oracle, k):
    """n - k <= 8: the parity comes straight from the Lagrange basis (one read of the data) — same bits as the transform pipeline
    (option encode_direct_max = 0) and as the oracle; out of place, in place, host memory."""
    torch = torch_cuda
    S = 24 if k < 50000 else 8
    x = np.random.default_rng(k).integers(0, P, size=(k, S), dtype=np.uint64).astype(np.uint32)
    N = 1 << max(1, int(np.ceil(np.log2(k))))
    for m in (1, 2, 3, 5, 8, 9):
        if m > N:
            continue
        want = expected(oracle, x, m)
        with fe.Encoder(k + m, k, 4 * S) as enc:
            for direct_max in (8, 0):
                enc.set_option("encode_direct_max", direct_max)
                out = torch_cuda.full((m * S,), 0x66666666, dtype=torch_cuda.int32, device="cuda:0")
                dx = to_dev(torch, x)
                enc.encode(dx, out)
                assert (to_host(out, (m, S)) == want).all(), (k, m, direct_max, enc.plan())
                assert (to_host(dx, (k, S)) == x).all()
                if m <= k:
                    enc.encode(dx)  # in place
                    got = to_host(dx, (k, S))
                    assert (got[:m] == want).all() and (got[m:] == x[m:]).all(), (k, m, direct_max)
                host_out = np.empty((m, S), dtype=np.uint32)
                enc.encode_host(x, host_out)
                assert (host_out == want).all(), (k, m, direct_max)
            with pytest.raises(fe.FastEccError):
                enc.set_option("encode_direct_max", 257)


def _few_loss_round_trips(torch, fe, enc, x, par, rng, counts):
    """Lose `e` blocks (at least one data block when possible), decode / repair on both decoder paths, compare with the originals."""
    k, S = x.shape
    m = par.shape[0]
    for e in counts:
        if e > m:
            continue
        lost = np.unique(np.r_[int(rng.integers(0, k)), rng.permutation(k + m)[: e - 1]])
        dp, pp = np.ones(k, np.uint8), np.ones(m, np.uint8)
        dp[lost[lost < k]] = 0
        pp[lost[lost >= k] - k] = 0
        bad_x, bad_p = x.copy(), par.copy()
        bad_x[dp == 0] = 0xA5A5A5A5
        bad_p[pp == 0] = 0x5A5A5A5A
        for direct_max in (16, 0):
            enc.set_option("decode_direct_max", direct_max)
            enc.decode_prepare(dp, pp)
            d, q = to_dev(torch, bad_x), to_dev(torch, bad_p)
            enc.decode(d, q)
            assert (to_host(d, (k, S)) == x).all(), (e, direct_max)
            assert (to_host(q, (m, S)) == bad_p).all(), (e, direct_max)
            enc.repair(d, q)
            assert (to_host(d, (k, S)) == x).all() and (to_host(q, (m, S)) == par).all(), (e, direct_max)
            hx, hp = bad_x.copy(), bad_p.copy()
            enc.repair(hx, hp, mem=fe.MEM_HOST)
            assert (hx == x).all() and (hp == par).all(), (e, direct_max)
        # parity only
        if m >= 2:
            dp, pp = np.ones(k, np.uint8), np.ones(m, np.uint8)
            pp[[0, m - 1]] = 0
            bad_p = par.copy()
            bad_p[pp == 0] = 7
            enc.set_option("decode_direct_max", 16)
            enc.decode_prepare(dp, pp)
            d, q = to_dev(torch, x), to_dev(torch, bad_p)
            enc.repair(d, q)
            assert (to_host(d, (k, S)) == x).all() and (to_host(q, (m, S)) == par).all()
    enc.set_option("decode_direct_max", 16)


@pytest.mark.parametrize("k,m", [(3, 2), (100, 30), (1000, 1000), (1024, 64), (3000, 700), (4096, 512), (5000, 5000), (20000, 3000)])
def test_few_losses_in_any_layout(torch_cuda, fe, oracle, k, m):
    """Zero-extended codes and codes with fewer parity blocks: up to 16 lost blocks are interpolated from the surviving data blocks
    plus as many parity blocks (no locator, no transform), the lost parity then re-evaluated from the data; same results as the
    transform path."""
    S = 24 if k < 10000 else 8
    rng = np.random.default_rng(k * 31 + m)
    x = rng.integers(0, P, size=(k, S), dtype=np.uint64).astype(np.uint32)
    par = expected(oracle, x, m)
    with fe.Encoder(k + m, k, 4 * S) as enc:
        _few_loss_round_trips(torch_cuda, fe, enc, x, par, rng, (1, 2, 3, 7, 16, 17))


@pytest.mark.parametrize("k,ratio", [(64, 4), (1024, 4), (512, 8)])
def test_few_losses_with_extra_cosets(torch_cuda, fe, k, ratio):
    """n = 4k, 8k: the parity sits on other cosets of the data points; the same interpolation."""
    torch = torch_cuda
    S = 16
    rng = np.random.default_rng(k + ratio)
    x = rng.integers(0, P, size=(k, S), dtype=np.uint64).astype(np.uint32)
    m = (ratio - 1) * k
    with fe.Encoder(ratio * k, k, 4 * S) as enc:
        out = torch.empty(m * S, dtype=torch.int32, device="cuda:0")
        enc.encode(to_dev(torch, x), out)
        par = to_host(out, (m, S)).copy()
        _few_loss_round_trips(torch, fe, enc, x, par, rng, (1, 4, 16))
