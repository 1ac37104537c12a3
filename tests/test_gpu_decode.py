"""GPU tests (-m gpu) of the erasure decoder (fastecc_decode_prepare / fastecc_decode, n = 2k).

The reference documents decoding (README.md:83-119, RS.md:42-79) and has no implementation, so there is nothing
upstream to pin against; the checks are (1) the size-independent round trip encode -> erase -> decode == original,
with the encoder itself pinned to the reference, and (2) the independent O(N^2) Lagrange interpolation of
oracle/fastecc_oracle.c (orc_decode).  Bit-exact."""
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


def pattern(rng, N, kind):
    """-> (data_present, parity_present) with at least N survivors"""
    dp, pp = np.ones(N, np.uint8), np.ones(N, np.uint8)
    if kind == "one":
        dp[rng.integers(N)] = 0
    elif kind == "all_data":
        dp[:] = 0
    elif kind == "half_each":
        dp[rng.permutation(N)[: N // 2]] = 0
        pp[rng.permutation(N)[: N - N // 2]] = 0
    elif kind == "random_max":  # exactly N erasures spread over the whole codeword
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


@pytest.mark.parametrize("logn", [1, 2, 3, 4, 5, 6, 7, 8, 10, 12])
@pytest.mark.parametrize("kind", ["one", "all_data", "half_each", "random_max", "quarter", "burst"])
def test_round_trip_and_lagrange(torch_cuda, fe, oracle, logn, kind):
    torch = torch_cuda
    N, S = 1 << logn, 77 if logn % 2 else 64
    rng = np.random.default_rng(logn * 100 + len(kind))
    x = rng.integers(0, P, size=(N, S), dtype=np.uint64).astype(np.uint32)
    par = oracle.encode_fast(x)
    dp, pp = pattern(rng, N, kind)
    damaged, dpar = x.copy(), par.copy()
    damaged[dp == 0] = 0xFFFFFFFF   # erased blocks hold garbage (not even field elements)
    dpar[pp == 0] = 0xDEADBEEF
    with fe.Encoder(2 * N, N, 4 * S) as enc:
        enc.decode_prepare(dp, pp)
        d = to_dev(torch, damaged)
        enc.decode(d, to_dev(torch, dpar))
        got = to_host(d, (N, S))
    assert (got == x).all()
    if N <= 256:
        assert (oracle.decode(damaged, dpar, dp, pp) == x).all()


def test_patterns_can_change_and_errors(torch_cuda, fe, oracle):
    torch = torch_cuda
    N, S = 64, 1025
    rng = np.random.default_rng(4)
    x = rng.integers(0, P, size=(N, S), dtype=np.uint64).astype(np.uint32)
    par = oracle.encode_fast(x)
    with fe.Encoder(2 * N, N, 4 * S) as enc:
        with pytest.raises(fe.FastEccError):       # no pattern yet
            enc.decode(to_dev(torch, x), to_dev(torch, par))
        dp, pp = np.ones(N, np.uint8), np.ones(N, np.uint8)
        dp[:40] = 0
        pp[:25] = 0                                 # 65 erasures > k
        with pytest.raises(fe.FastEccError) as ei:
            enc.decode_prepare(dp, pp)
        assert ei.value.code == fe.E_INVAL
        for seed in range(3):                       # the same context serves pattern after pattern
            dp, pp = pattern(np.random.default_rng(seed), N, "random_max")
            damaged = x.copy()
            damaged[dp == 0] = 0
            enc.decode_prepare(dp, pp)
            d = to_dev(torch, damaged)
            enc.decode(d, to_dev(torch, par))
            assert (to_host(d, (N, S)) == x).all()
            host = damaged.copy()                   # host-memory form
            enc.decode(host, par, mem=fe.MEM_HOST)
            assert (host == x).all()
        dp, pp = np.ones(N, np.uint8), np.ones(N, np.uint8)
        pp[::2] = 0                                 # only parity lost: nothing to do, data untouched
        enc.decode_prepare(dp, pp)
        d = to_dev(torch, x)
        enc.decode(d, to_dev(torch, par))
        assert (to_host(d, (N, S)) == x).all()
    with fe.Encoder(2 * 96, 96, 64, flags=fe.CODE_MIXED_RADIX) as enc:   # the mixed-radix codes decode too (tests/test_gpu_mixed.py)
        enc.decode_prepare(np.ones(96, np.uint8), np.ones(96, np.uint8))
        with pytest.raises(fe.FastEccError) as ei:                        # and refuse what no code can repair
            enc.decode_prepare(np.zeros(96, np.uint8), np.r_[np.zeros(1, np.uint8), np.ones(95, np.uint8)])
        assert ei.value.code == fe.E_INVAL


@pytest.mark.parametrize("lost_fraction", [0.02, 0.5])
def test_headline_size_round_trip(torch_cuda, fe, lost_fraction):
    """(n,k) = (2^20, 2^19), 4 KB blocks: encode, lose blocks all over the codeword, decode, compare on the device."""
    torch = torch_cuda
    N, S = 1 << 19, 1024
    g = torch.Generator(device="cuda:0").manual_seed(11)
    data = torch.randint(0, P, (N * S,), dtype=torch.int64, device="cuda:0", generator=g).to(torch.int32)
    parity = torch.empty_like(data)
    rng = np.random.default_rng(int(lost_fraction * 100))
    lost = rng.permutation(2 * N)[: int(2 * N * lost_fraction)]
    dp, pp = np.ones(N, np.uint8), np.ones(N, np.uint8)
    dp[lost[lost < N]] = 0
    pp[lost[lost >= N] - N] = 0
    with fe.Encoder(2 * N, N, 4 * S) as enc:
        enc.encode(data, parity)
        damaged = data.clone().view(N, S)
        damaged[torch.from_numpy(dp == 0).to("cuda:0")] = -1
        parity.view(N, S)[torch.from_numpy(pp == 0).to("cuda:0")] = 0x5A5A5A5A
        enc.decode_prepare(dp, pp)
        enc.decode(damaged, parity)
        torch.cuda.synchronize()
        assert bool((damaged.view(-1) == data).all())


@pytest.mark.parametrize("logn,S", [(18, 8), (18, 7), (19, 4), (18, 66), (17, 16), (17, 5)])
def test_split_transform_matches_the_2k_point_transform(torch_cuda, fe, logn, S):
    """(2k,k) codes with k >= 2^17 decode through two half-size transforms (option "decode_split", default on): the data half and the few
    parity block groups in use each go through the first DIF tile with per-block factors, the parity half's coefficients join the data half's
    between the two halves of the MID tile.  Same bits as the single 2k-point transform and as the original stripe; patterns that need
    fewer parity groups than the one before (the zeroing of the groups no longer written), ragged and odd block sizes, repair."""
    torch = torch_cuda
    N = 1 << logn
    g = torch.Generator(device="cuda:0").manual_seed(logn * 100 + S)
    data = torch.randint(0, P, (N * S,), dtype=torch.int64, device="cuda:0", generator=g).to(torch.int32)
    parity = torch.empty_like(data)
    rng = np.random.default_rng(S)
    with fe.Encoder(2 * N, N, 4 * S) as enc:
        enc.encode(data, parity)
        for count in (N // 2, 300, N, 2 * N // 50, 257):  # many parity groups in use, then few, then all of them, ...
            lost = rng.permutation(2 * N)[:count]
            if count == N:
                lost = np.concatenate([np.arange(N // 2), N + rng.permutation(N)[: N // 2]])  # half of the data, half of the parity
            dp, pp = np.ones(N, np.uint8), np.ones(N, np.uint8)
            dp[lost[lost < N]] = 0
            pp[lost[lost >= N] - N] = 0
            results = []
            for split in (1, 2, 0):  # 1: the small form where the pattern allows it (else block groups), 2: block groups only, 0: one 2k-point transform
                enc.set_option("decode_split", split)
                enc.decode_prepare(dp, pp)
                damaged, dpar = data.clone(), parity.clone()
                damaged.view(N, S)[torch.from_numpy(dp == 0).to("cuda:0")] = -1
                dpar.view(N, S)[torch.from_numpy(pp == 0).to("cuda:0")] = 0x5A5A5A5A
                enc.profile(True)
                enc.profile_reset()
                enc.decode(damaged, dpar)
                torch.cuda.synchronize()
                prof = enc.profile_read()
                enc.profile(False)
                assert ("decode_split_transform" in prof) == (split != 0) and ("decode_transform_2k" in prof) == (split == 0), prof  # which path ran
                # which FORM of the split transform ran shows in the bytes the step accounts for: the small form moves the three passes over the
                # data half plus three over the k >> 5 rows of the parity half in use (few losses: every 32nd parity block is enough)
                if split == 1 and count in (300, 257, 2 * N // 50):
                    assert prof["decode_split_transform"][2] == (3 * N + 3 * (N >> 5)) * 4 * S, prof
                if split == 2:
                    assert prof["decode_split_transform"][2] != (3 * N + 3 * (N >> 5)) * 4 * S, prof
                results.append(damaged)
                assert bool((damaged == data).all()), (count, split)
                # repair from the damaged stripes again: the lost parity blocks come from a second chain over the same two half transforms
                damaged2, dpar2 = data.clone(), parity.clone()
                damaged2.view(N, S)[torch.from_numpy(dp == 0).to("cuda:0")] = -3
                dpar2.view(N, S)[torch.from_numpy(pp == 0).to("cuda:0")] = 0x5A5A5A5A
                enc.profile(True)
                enc.profile_reset()
                enc.repair(damaged2, dpar2)
                torch.cuda.synchronize()
                prof = enc.profile_read()
                enc.profile(False)
                if (pp == 0).any():
                    assert ("repair_split_transform" in prof) == (split != 0), prof
                assert bool((damaged2 == data).all()) and bool((dpar2 == parity).all()), (count, split)
            assert torch.equal(results[0], results[1]) and torch.equal(results[0], results[2])
        enc.set_option("decode_split", 1)
        with pytest.raises(fe.FastEccError):
            enc.set_option("decode_split", 3)


@pytest.mark.parametrize("count", [3, 200, 5000, 1 << 16])
def test_host_stripes_only_the_rebuilt_blocks_travel_back(torch_cuda, fe, count):
    """FASTECC_MEM_HOST decode / repair at k = 2^17: the stripes are staged through HBM (of the parity stripe only the block groups the split
    transform reads), and with few enough lost blocks only those come back — packed on the device, one copy, a memcpy per block.  The surviving
    blocks of the host stripes must be untouched (they hold a marker the device never saw changed), the lost ones restored."""
    torch = torch_cuda
    N, S = 1 << 17, 24
    g = torch.Generator(device="cuda:0").manual_seed(count)
    data = torch.randint(0, P, (N * S,), dtype=torch.int64, device="cuda:0", generator=g).to(torch.int32)
    parity = torch.empty_like(data)
    rng = np.random.default_rng(count)
    lost = rng.permutation(2 * N)[:count]
    dp, pp = np.ones(N, np.uint8), np.ones(N, np.uint8)
    dp[lost[lost < N]] = 0
    pp[lost[lost >= N] - N] = 0
    with fe.Encoder(2 * N, N, 4 * S) as enc:
        enc.encode(data, parity)
        torch.cuda.synchronize()
        x, par = to_host(data, (N, S)).copy(), to_host(parity, (N, S)).copy()
        enc.decode_prepare(dp, pp)
        for repair in (False, True, True):  # (the second repair: lists and staging buffers of the first are reused)
            hd, hq = x.copy(), par.copy()
            hd[dp == 0] = 0xABABABAB
            hq[pp == 0] = 0xCDCDCDCD
            if repair:
                enc.repair(hd, hq, mem=fe.MEM_HOST)
                assert np.array_equal(hd, x) and np.array_equal(hq, par), (count, repair)
            else:
                enc.decode(hd, hq, mem=fe.MEM_HOST)
                want_q = par.copy()
                want_q[pp == 0] = 0xCDCDCDCD
                assert np.array_equal(hd, x) and np.array_equal(hq, want_q), (count, repair)
        # another pattern on the same context: the row lists follow it
        dp2, pp2 = np.ones(N, np.uint8), np.ones(N, np.uint8)
        dp2[[5, 77, N - 1]] = 0
        pp2[[0]] = 0
        enc.decode_prepare(dp2, pp2)
        hd, hq = x.copy(), par.copy()
        hd[dp2 == 0] = 1
        hq[pp2 == 0] = 2
        enc.repair(hd, hq, mem=fe.MEM_HOST)
        assert np.array_equal(hd, x) and np.array_equal(hq, par)
        # pinned host memory is accepted as well (the same staging)
        td, tq = torch.from_numpy(x.view(np.int32).copy()).pin_memory(), torch.from_numpy(par.view(np.int32).copy()).pin_memory()
        td.view(N, S)[torch.from_numpy(dp2 == 0)] = 3
        tq.view(N, S)[torch.from_numpy(pp2 == 0)] = 4
        enc.repair(td, tq, mem=fe.MEM_HOST_PINNED)
        assert np.array_equal(td.numpy().view(np.uint32).reshape(N, S), x) and np.array_equal(tq.numpy().view(np.uint32).reshape(N, S), par)


@pytest.mark.parametrize("logn,tag", [(18, "SW32:"), (19, "SW4x32:")])
def test_split_transform_on_tiles_of_several_windows(torch_cuda, fe, logn, tag):
    """16 KB blocks at k = 2^18 / 2^19: the outer tiles of the plan address their blocks through two / four windows (a tile spans 4 / 8 GiB);
    the per-block-factor modes of the split transform exist for those tiles too.  One 2 % pattern, decode and repair, both forms of the transform."""
    torch = torch_cuda
    N, S = 1 << logn, 4096
    g = torch.Generator(device="cuda:0").manual_seed(5)
    data = torch.randint(0, 2**31 - 1, (N * S,), dtype=torch.int32, device="cuda:0", generator=g)   # (words below 2^31 < p: no 64-bit detour for 4 GiB)
    parity = torch.empty_like(data)
    rng = np.random.default_rng(5)
    lost = rng.permutation(2 * N)[: 2 * N // 50]
    dp, pp = np.ones(N, np.uint8), np.ones(N, np.uint8)
    dp[lost[lost < N]] = 0
    pp[lost[lost >= N] - N] = 0
    di, pi = torch.from_numpy(np.flatnonzero(dp == 0)).to("cuda:0"), torch.from_numpy(np.flatnonzero(pp == 0)).to("cuda:0")
    with fe.Encoder(2 * N, N, 4 * S) as enc:
        assert tag in enc.plan(), enc.plan()
        enc.encode(data, parity)
        saved_d, saved_p = data.view(N, S)[di].clone(), parity.view(N, S)[pi].clone()
        for split in (1, 0):
            enc.set_option("decode_split", split)
            enc.decode_prepare(dp, pp)
            data.view(N, S)[di] = -1
            parity.view(N, S)[pi] = -2
            enc.profile(True)
            enc.profile_reset()
            enc.decode(data, parity)
            torch.cuda.synchronize()
            prof = enc.profile_read()
            enc.profile(False)
            assert ("decode_split_transform" in prof) == (split == 1), prof
            assert torch.equal(data.view(N, S)[di], saved_d)
            data.view(N, S)[di] = -3
            enc.repair(data, parity)
            torch.cuda.synchronize()
            assert torch.equal(data.view(N, S)[di], saved_d) and torch.equal(parity.view(N, S)[pi], saved_p)
        enc.set_option("decode_split", 1)


@pytest.mark.parametrize("n,k,S", [(600000, 300000, 4), (2 * 262145, 262145, 6), (300000 + 270000, 300000, 8),
                                   ((1 << 18) + (1 << 17), 1 << 18, 8), ((1 << 18) + 100000, 1 << 18, 5), (300000 + 140000, 300000, 4), ((1 << 18) + (1 << 14), 1 << 18, 16)])
def test_split_transform_of_zero_extended_codes(torch_cuda, fe, n, k, S):
    """k and n - k between 2^18 and 2^19, no powers of two: the code lives inside the (2^20, 2^19) code (zero extension, fold 0), its stripes
    hold fewer than 2^19 blocks — the split transform reads them with a bound, the blocks beyond count as zero (data) or lost (parity).
    Codes with fewer parity blocks (n - k <= N / 2: parity block j sits at block j << fold of the parity half, fold = 1 ... 4) have the parity
    blocks in use copied to their places first."""
    torch = torch_cuda
    m = n - k
    g = torch.Generator(device="cuda:0").manual_seed(k % 1000 + S)
    data = torch.randint(0, P, (k * S,), dtype=torch.int64, device="cuda:0", generator=g).to(torch.int32)
    parity = torch.empty(m * S, dtype=torch.int32, device="cuda:0")
    rng = np.random.default_rng(S)
    with fe.Encoder(n, k, 4 * S) as enc:
        enc.encode(data, parity)
        for count in (min(n // 40, m), m, 1000):
            lost = rng.permutation(n)[:count]
            dp, pp = np.ones(k, np.uint8), np.ones(m, np.uint8)
            dp[lost[lost < k]] = 0
            pp[lost[lost >= k] - k] = 0
            for split in (1, 0):
                enc.set_option("decode_split", split)
                enc.decode_prepare(dp, pp)
                damaged, dpar = data.clone(), parity.clone()
                damaged.view(k, S)[torch.from_numpy(dp == 0).to("cuda:0")] = -1
                dpar.view(m, S)[torch.from_numpy(pp == 0).to("cuda:0")] = 0x5A5A5A5A
                enc.profile(True)
                enc.profile_reset()
                enc.decode(damaged, dpar)
                torch.cuda.synchronize()
                prof = enc.profile_read()
                enc.profile(False)
                assert ("decode_split_transform" in prof) == (split == 1), prof
                assert bool((damaged == data).all()), (count, split)
                damaged.view(k, S)[torch.from_numpy(dp == 0).to("cuda:0")] = -7   # repair from the damaged stripes
                enc.profile(True)
                enc.profile_reset()
                enc.repair(damaged, dpar)
                torch.cuda.synchronize()
                prof = enc.profile_read()
                enc.profile(False)
                assert bool((damaged == data).all()) and bool((dpar == parity).all()), (count, split)
                # fold 0 (n - k > N / 2): the lost parity blocks come from the split transform's second chain; fewer parity blocks: a second encode
                n_pow = 1 << (k - 1).bit_length()
                if (pp == 0).any():
                    assert ("repair_split_transform" in prof) == (split == 1 and 2 * m > n_pow), prof
        enc.set_option("decode_split", 1)


def test_sector_pipeline_pack_encode_lose_decode_unpack(torch_cuda, fe):
    """README.md:160-163 end to end: arbitrary 4096-byte sectors -> 4100-byte blocks -> parity; lose 30 % of the
    codeword; decode; unpack; the sectors come back bit for bit."""
    torch = torch_cuda
    k, W = 1 << 12, 1024
    g = torch.Generator(device="cuda:0").manual_seed(77)
    sectors = torch.randint(-(1 << 31), 1 << 31, (k * W,), dtype=torch.int64, device="cuda:0", generator=g).to(torch.int32)
    sectors.view(k, W)[::7, ::3] |= -1048576   # 0xFFF00000: plenty of digits that need recoding
    rng = np.random.default_rng(5)
    lost = rng.permutation(2 * k)[: int(0.3 * 2 * k)]
    dp, pp = np.ones(k, np.uint8), np.ones(k, np.uint8)
    dp[lost[lost < k]] = 0
    pp[lost[lost >= k] - k] = 0
    with fe.Encoder(2 * k, k, 4 * (W + 1)) as enc:
        blocks = torch.empty(k * (W + 1), dtype=torch.int32, device="cuda:0")
        parity = torch.empty_like(blocks)
        enc.pack_blocks(sectors, blocks)
        assert enc.check_range(blocks) == 0
        enc.encode(blocks, parity)
        blocks.view(k, W + 1)[torch.from_numpy(dp == 0).to("cuda:0")] = 0x7BADF00D
        parity.view(k, W + 1)[torch.from_numpy(pp == 0).to("cuda:0")] = -1
        enc.decode_prepare(dp, pp)
        enc.decode(blocks, parity)
        back = torch.empty_like(sectors)
        assert enc.unpack_blocks(blocks, back) == 0
        assert bool((back == sectors).all())


@pytest.mark.parametrize("logn,d", [(6, 1), (10, 2), (12, 1), (12, 4), (14, 3)])
def test_codes_with_fewer_parity_blocks(torch_cuda, fe, oracle, logn, d):
    """(k + k/2^d, k): decoded inside the (2k,k) code, the unused parity positions counting as erased."""
    torch = torch_cuda
    N, S = 1 << logn, 48
    M = N >> d
    rng = np.random.default_rng(logn * 10 + d)
    x = rng.integers(0, P, size=(N, S), dtype=np.uint64).astype(np.uint32)
    par = oracle.encode_fast(x)[:: 1 << d]
    lost = rng.permutation(N + M)[:M]            # as many erasures as the code can take
    dp, pp = np.ones(N, np.uint8), np.ones(M, np.uint8)
    dp[lost[lost < N]] = 0
    pp[lost[lost >= N] - N] = 0
    damaged = x.copy()
    damaged[dp == 0] = 0
    with fe.Encoder(N + M, N, 4 * S) as enc:
        enc.decode_prepare(dp, pp)
        dd = to_dev(torch, damaged)
        enc.decode(dd, to_dev(torch, par))
        assert (to_host(dd, (N, S)) == x).all()


@pytest.mark.parametrize("logn,e", [(1, 2), (5, 2), (6, 3), (10, 2), (11, 3), (13, 2)])
def test_codes_with_more_parity_blocks(torch_cuda, fe, oracle, logn, e):
    """n = 4k, 8k: decoded on the n-th roots of unity; up to n - k blocks may be lost, e.g. ALL data and most parity."""
    torch = torch_cuda
    N, S = 1 << logn, 40
    M = ((1 << e) - 1) * N
    rng = np.random.default_rng(logn * 10 + e)
    x = rng.integers(0, P, size=(N, S), dtype=np.uint64).astype(np.uint32)
    with fe.Encoder(N << e, N, 4 * S) as enc:
        dx = to_dev(torch, x)
        par = torch.empty(M * S, dtype=torch.int32, device="cuda:0")
        enc.encode(dx, par)
        for trial in range(2):
            lost = rng.permutation(N + M)[:M] if trial == 0 else np.concatenate([np.arange(N), N + rng.permutation(M)[: M - N]])
            dp, pp = np.ones(N, np.uint8), np.ones(M, np.uint8)
            dp[lost[lost < N]] = 0
            pp[lost[lost >= N] - N] = 0
            damaged = x.copy()
            damaged[dp == 0] = 0xFFFFFFFF
            dpar = par.clone()
            dpar.view(M, S)[torch.from_numpy(pp == 0).to("cuda:0")] = 0x0BADBEEF
            enc.decode_prepare(dp, pp)
            dd = to_dev(torch, damaged)
            enc.decode(dd, dpar)
            assert (to_host(dd, (N, S)) == x).all(), (logn, e, trial)
        dp[:] = 0
        pp[:] = 0
        pp[: N - 1] = 1                      # one block short of k survivors
        if N > 1:
            with pytest.raises(fe.FastEccError) as ei:
                enc.decode_prepare(dp, pp)
            assert ei.value.code == fe.E_INVAL


# ------------------------------------------------------------------------------------------------
# fastecc_repair: the lost parity blocks come back too
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n_over_k,k,S", [(2.0, 1 << 10, 64), (2.0, 1 << 13, 33), (1.25, 1 << 8, 64), (4.0, 1 << 6, 16), (1.2, 1000, 24)])
def test_repair_restores_data_and_parity(torch_cuda, fe, oracle, n_over_k, k, S):
    torch = torch_cuda
    n = int(k * n_over_k)
    m = n - k
    rng = np.random.default_rng(k + m)
    x = rng.integers(0, P, size=(k, S), dtype=np.uint64).astype(np.uint32)
    with fe.Encoder(n, k, 4 * S) as enc:
        d = to_dev(torch, x)
        par_dev = torch.empty(m * S, dtype=torch.int32, device="cuda:0")
        enc.encode(d, par_dev)  # the codeword by the (pinned) encoder itself: every code family has its own parity
        torch.cuda.synchronize()
        par = to_host(par_dev, (m, S))
        lost = rng.permutation(n)[:m]  # as many losses as the code tolerates, over data and parity
        dp, pp = np.ones(k, np.uint8), np.ones(m, np.uint8)
        dp[lost[lost < k]] = 0
        pp[lost[lost >= k] - k] = 0
        damaged, dpar = x.copy(), par.copy()
        damaged[dp == 0] = 0xFFFFFFFF
        dpar[pp == 0] = 0xDEADBEEF
        enc.decode_prepare(dp, pp)
        dd, dq = to_dev(torch, damaged), to_dev(torch, dpar)
        enc.repair(dd, dq)
        torch.cuda.synchronize()
        assert (to_host(dd, (k, S)) == x).all()
        assert (to_host(dq, (m, S)) == par).all()
        # host memory form
        hd, hp = damaged.copy(), dpar.copy()
        enc.repair(hd, hp, mem=fe.MEM_HOST)
        assert (hd == x).all() and (hp == par).all()
        # only parity lost: nothing to decode, the parity is simply encoded again
        dp2, pp2 = np.ones(k, np.uint8), np.ones(m, np.uint8)
        pp2[: max(1, m // 3)] = 0
        enc.decode_prepare(dp2, pp2)
        dq2 = to_dev(torch, np.where(pp2[:, None] == 0, np.uint32(7), par))
        enc.repair(to_dev(torch, x), dq2)
        torch.cuda.synchronize()
        assert (to_host(dq2, (m, S)) == par).all()
        # plain decode leaves the parity alone
        enc.decode_prepare(dp, pp)
        dq3 = to_dev(torch, dpar)
        enc.decode(to_dev(torch, damaged), dq3)
        torch.cuda.synchronize()
        assert (to_host(dq3, (m, S)) == dpar).all()


@pytest.mark.parametrize("N,S", [(2, 5), (16, 64), (2048, 37), (1 << 15, 8)])
def test_few_losses_take_the_direct_path(torch_cuda, fe, oracle, N, S):
    """Up to 16 lost blocks of a (2k,k) codeword are recomputed straight from the survivors (no locator tree, no transform):
    same results as the transform path (option decode_direct_max = 0), as the original data and parity, and — small N — as the
    oracle's Lagrange decoder; data only (decode) and data + parity (repair), device and host stripes."""
    torch = torch_cuda
    rng = np.random.default_rng(N + S)
    x = rng.integers(0, P, size=(N, S), dtype=np.uint64).astype(np.uint32)
    par = oracle.encode_fast(x)
    with fe.Encoder(2 * N, N, 4 * S) as enc:
        for e in (1, 2, 3, 5, 8, 16, 17):
            if e > N:
                continue
            lost = rng.permutation(2 * N)[:e]
            lost[0] = 2 * int(rng.integers(0, N)) // 2  # a data block for sure
            lost = np.unique(np.r_[lost[0] % N, lost[1:]])
            dp, pp = np.ones(N, np.uint8), np.ones(N, np.uint8)
            dp[lost[lost < N]] = 0
            pp[lost[lost >= N] - N] = 0
            bad_x, bad_p = x.copy(), par.copy()
            bad_x[dp == 0] = 0xA5A5A5A5
            bad_p[pp == 0] = 0x5A5A5A5A
            results = []
            for direct_max in (16, 0):
                enc.set_option("decode_direct_max", direct_max)
                enc.decode_prepare(dp, pp)
                d, q = to_dev(torch, bad_x), to_dev(torch, bad_p)
                enc.decode(d, q)
                assert (to_host(d, (N, S)) == x).all(), (e, direct_max)
                assert (to_host(q, (N, S)) == bad_p).all(), (e, direct_max)  # decode leaves the parity alone
                enc.repair(d, q)
                assert (to_host(q, (N, S)) == par).all(), (e, direct_max)
                hx, hp = bad_x.copy(), bad_p.copy()
                enc.repair(hx, hp, mem=fe.MEM_HOST)
                assert (hx == x).all() and (hp == par).all(), (e, direct_max)
                results.append(to_host(d, (N, S)).copy())
            assert (results[0] == results[1]).all()
            if N <= 256:
                assert (oracle.decode(bad_x, bad_p, dp, pp) == x).all()
        enc.set_option("decode_direct_max", 16)
        with pytest.raises(fe.FastEccError):
            enc.set_option("decode_direct_max", 257)
