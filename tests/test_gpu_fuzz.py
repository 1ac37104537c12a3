"""GPU differential test (-m gpu): seeded random configurations of everything the encode / decode entry points accept —
(n, k) of any shape, ragged block sizes, buffers at odd word offsets, every kind of plan, device / host / block-pointer
forms, mixed-radix orders, the 64-bit field — each checked bit for bit against the oracle, then damaged with a random
erasure pattern and repaired.  The fixed grids of the other test files pick their corners by hand; this one does not."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

P = 0xFFF00001
P61 = (1 << 61) - 1
PLANS = (0, 51, 52, 54, 42, 34, 1100, 2080, 3100, 1090, 3090, 4090)


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    return torch


@pytest.fixture(scope="module")
def fe(hip_lib):
    import fastecc_amd
    return fastecc_amd


def expected_pow2(oracle, x, m):
    """fastecc_create's code for any (k, m <= N): the (N + M, N) sub-code of the zero-extended stripe (RS.md:23-33)."""
    k, S = x.shape
    lg = max(1, int(np.ceil(np.log2(k))))
    N = 1 << lg
    lgm = int(np.ceil(np.log2(m))) if m > 1 else 0
    fold = min(lg - lgm, 4)
    padded = np.zeros((N, S), dtype=np.uint32)
    padded[:k] = x
    return oracle.encode_fast(padded)[:: 1 << fold][:m]


def offset_view(torch, a, off):
    """The array on the device, starting `off` words into its allocation (4-byte alignment is all the ABI asks for)."""
    flat = np.ascontiguousarray(a).view(np.int32).reshape(-1)
    buf = torch.empty(flat.size + off, dtype=torch.int32, device="cuda:0")
    view = buf[off:]
    view.copy_(torch.from_numpy(flat))
    return view


def random_case(rng):
    kind = rng.choice(["pow2", "any", "any", "mixed", "mixed"])
    if kind == "pow2":
        k = 1 << int(rng.integers(1, 13))
        m = k
    elif kind == "any":
        k = int(rng.integers(1, 3000))
        N = 1 << max(1, int(np.ceil(np.log2(k))))
        m = int(rng.integers(1, N + 1))
    else:
        k = int(rng.integers(3, 3000))
        m = 0
    S = int(rng.choice([1, 2, 3, 5, 8, 17, 32, 33, 64, 100, 128, 255, 256, 320]))
    if k * S > 400000:
        S = max(1, 400000 // k)
    return kind, k, m, S


@pytest.mark.parametrize("seed", range(240))
def test_random_u32_configuration(torch_cuda, fe, oracle, seed):
    torch = torch_cuda
    rng = np.random.default_rng(1000 + seed)
    kind, k, m, S = random_case(rng)
    flags = 0
    if kind == "mixed":
        flags = fe.CODE_MIXED_RADIX
        order = fe.mixed_radix_order(k)
        m = int(rng.integers(1, order + 1))
        if not order & (order - 1):
            kind = "any"  # a power of two: the ordinary code
    x = rng.integers(0, P, size=(k, S), dtype=np.uint64).astype(np.uint32)
    x.reshape(-1)[rng.integers(0, x.size, size=min(4, x.size))] = [0, 1, P - 1, 0x000FFFFF][: min(4, x.size)]
    want = oracle.encode_mixed_code(x, k + m, order) if kind == "mixed" else expected_pow2(oracle, x, m)
    what = (kind, k, m, S)
    with fe.Encoder(k + m, k, 4 * S, flags=flags) as enc:
        plan = int(rng.choice(PLANS))
        try:
            enc.set_plan(plan)
        except fe.FastEccError:
            plan = 0
        what += (plan, enc.plan())
        off_d, off_p = int(rng.integers(0, 4)), int(rng.integers(0, 4))
        d = offset_view(torch, x, off_d)
        out = offset_view(torch, np.full((m, S), 0x55555555, dtype=np.uint32), off_p)
        enc.encode(d, out)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy().view(np.uint32).reshape(m, S), want), what
        assert np.array_equal(d.cpu().numpy().view(np.uint32).reshape(k, S), x), what
        form = rng.choice(["host", "inplace", "blocks", "none"])
        if form == "host":
            got = np.empty((m, S), dtype=np.uint32)
            enc.encode_host(x, got)
            assert np.array_equal(got, want), what
        elif form == "inplace" and m <= k:
            d2 = offset_view(torch, x, off_p)
            enc.encode(d2)
            got = d2.cpu().numpy().view(np.uint32).reshape(k, S)
            assert np.array_equal(got[:m], want) and np.array_equal(got[m:], x[m:]), what
        elif form == "blocks" and m <= k:
            blocks = [np.ascontiguousarray(x[i]).copy() for i in range(k)]
            enc.encode_blocks([b.ctypes.data for b in blocks])
            assert np.array_equal(np.stack(blocks[:m]), want), what
        # erasures: up to m of the k + m blocks, anywhere
        lost = rng.permutation(k + m)[: int(rng.integers(1, m + 1))]
        dp, pp = np.ones(k, np.uint8), np.ones(m, np.uint8)
        dp[lost[lost < k]] = 0
        pp[lost[lost >= k] - k] = 0
        bad_x, bad_p = x.copy(), want.copy()
        bad_x[dp == 0] = 0xABABABAB
        bad_p[pp == 0] = 0xCDCDCDCD
        enc.decode_prepare(dp, pp)
        if rng.integers(0, 2):
            dd, dq = offset_view(torch, bad_x, 0), offset_view(torch, bad_p, 0)
            enc.repair(dd, dq)
            torch.cuda.synchronize()
            assert np.array_equal(dd.cpu().numpy().view(np.uint32).reshape(k, S), x), what
            assert np.array_equal(dq.cpu().numpy().view(np.uint32).reshape(m, S), want), what
        else:
            enc.repair(bad_x, bad_p, mem=fe.MEM_HOST)
            assert np.array_equal(bad_x, x) and np.array_equal(bad_p, want), what


@pytest.mark.parametrize("seed", range(48))
def test_random_p61_configuration(torch_cuda, fe, seed):
    from oracle import OracleP61
    orc = OracleP61()
    torch = torch_cuda
    rng = np.random.default_rng(5000 + seed)
    k = 1 << int(rng.integers(1, 12))
    E = int(rng.choice([1, 2, 3, 4, 7, 16, 33, 64, 65]))  # 16-byte elements per block
    if k * E > 60000:
        E = max(1, 60000 // k)
    x = rng.integers(0, P61, size=(k, 2 * E), dtype=np.uint64)
    x.reshape(-1)[:2] = [0, P61 - 1]
    want = orc.encode(x)
    with fe.Encoder(2 * k, k, 16 * E, field=fe.FIELD_GF_P61_SQUARED) as enc:
        plan = int(rng.choice([0, 1, 2, 3, 4, 12, 13, 14, 23, 24]))
        try:
            enc.set_plan(plan)
        except fe.FastEccError:
            plan = 0
        what = (k, E, plan, enc.plan())
        d = torch.from_numpy(x.view(np.int64).reshape(-1).copy()).to("cuda:0")
        out = torch.empty_like(d)
        enc.encode(d, out)
        torch.cuda.synchronize()
        par = out.cpu().numpy().view(np.uint64).reshape(k, 2 * E)
        assert np.array_equal(par, want), what
        lost = rng.permutation(2 * k)[: int(rng.integers(1, k + 1))]
        dp, pp = np.ones(k, np.uint8), np.ones(k, np.uint8)
        dp[lost[lost < k]] = 0
        pp[lost[lost >= k] - k] = 0
        bx, bp = x.copy(), want.copy()
        bx[dp == 0] = 7
        bp[pp == 0] = 9
        enc.decode_prepare(dp, pp)
        if seed % 2:
            enc.repair(bx, bp, mem=fe.MEM_HOST)  # stripes in host memory
            assert np.array_equal(bx, x) and np.array_equal(bp, want), what
            return
        dd = torch.from_numpy(bx.view(np.int64).reshape(-1)).to("cuda:0")
        dq = torch.from_numpy(bp.view(np.int64).reshape(-1)).to("cuda:0")
        enc.repair(dd, dq)
        torch.cuda.synchronize()
        assert np.array_equal(dd.cpu().numpy().view(np.uint64).reshape(k, 2 * E), x), what
        assert np.array_equal(dq.cpu().numpy().view(np.uint64).reshape(k, 2 * E), want), what


@pytest.mark.parametrize("seed", range(40))
def test_random_sharded_batched_and_column_calls(torch_cuda, fe, oracle, seed):
    """The (2k,k) code through the three other ways in: one stripe in column slabs (all slabs on this GPU), column ranges
    of one context, and many stripes per launch."""
    torch = torch_cuda
    rng = np.random.default_rng(9000 + seed)
    k = 1 << int(rng.integers(1, 12))
    G = int(rng.choice([1, 2, 4, 8]))
    w = int(rng.choice([1, 3, 8, 32, 64]))  # words per slab
    S = G * w
    if k * S > 300000:
        k = max(2, 1 << int(np.log2(300000 // S)))
    x = rng.integers(0, P, size=(k, S), dtype=np.uint64).astype(np.uint32)
    want = oracle.encode_fast(x)
    to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32).reshape(-1).copy()).to("cuda:0")  # noqa: E731
    host = lambda t, rows, cols: t.cpu().numpy().view(np.uint32).reshape(rows, cols)  # noqa: E731
    enc = fe.ShardedEncoder(2 * k, k, 4 * S, [0] * G)
    try:
        enc.set_option("sub_slabs", int(rng.choice([1, 2, 4, 8])))
        enc.set_option("gather_mode", int(rng.choice([1, 2])))
        what = (k, G, w, enc.plan())
        full = to_dev(x)
        out = torch.empty_like(full)
        enc.encode(full, out)
        torch.cuda.synchronize()
        assert np.array_equal(host(out, k, S), want), what
        slabs = [to_dev(x[:, g * w:(g + 1) * w]) for g in range(G)]
        pslabs = [torch.empty_like(t) for t in slabs]
        gathered = torch.zeros_like(full)
        enc.encode_sharded(slabs, pslabs, gathered)
        torch.cuda.synchronize()
        assert np.array_equal(host(gathered, k, S), want), what
        for g in range(G):
            assert np.array_equal(host(pslabs[g], k, w), want[:, g * w:(g + 1) * w]), what
        hout = np.empty_like(x)
        enc.encode(x, hout, mem=fe.MEM_HOST)
        assert np.array_equal(hout, want), what
    finally:
        enc.close()
    with fe.Encoder(2 * k, k, 4 * S) as one:
        # column ranges in multiples of 32 words where the block has them, else the whole block
        d = to_dev(x)
        out = torch.zeros_like(d)
        if S % 32 == 0:
            cut = 32 * int(rng.integers(0, S // 32 + 1))
            if cut:
                one.encode_columns(d, out, 0, cut)
            if cut < S:
                one.encode_columns(d, out, cut, S - cut)
        else:
            one.encode_columns(d, out, 0, S)
        torch.cuda.synchronize()
        assert np.array_equal(host(out, k, S), want), (k, S, one.plan())
        count = int(rng.integers(1, 6))
        many = rng.integers(0, P, size=(count, k, S), dtype=np.uint64).astype(np.uint32)
        many[0] = x
        dm = to_dev(many)
        om = torch.empty_like(dm)
        one.encode_batch(dm, om, count)
        torch.cuda.synchronize()
        got = om.cpu().numpy().view(np.uint32).reshape(count, k, S)
        for i in range(count):
            assert np.array_equal(got[i], oracle.encode_fast(many[i])), (k, S, count, i, one.plan())


@pytest.mark.parametrize("seed", range(40))
def test_random_large_mixed_radix_configuration(torch_cuda, fe, oracle, seed):
    """Mixed-radix orders large enough for the fused odd-radix tiles (2^m with m >= 11), random k (zero extension) and parity counts
    (truncation), narrow blocks; encode against the oracle, then lose up to 20 blocks and repair (both decoder paths occur)."""
    torch = torch_cuda
    rng = np.random.default_rng(7000 + seed)
    k = int(rng.integers(3 << 11, 120000))
    order = fe.mixed_radix_order(k)
    if not order & (order - 1):
        pytest.skip("a power of two")
    m = int(rng.integers(1, order + 1)) if seed % 3 else int(rng.integers(1, 20))
    S = int(rng.choice([1, 2, 3, 4]))
    x = rng.integers(0, P, size=(k, S), dtype=np.uint64).astype(np.uint32)
    want = oracle.encode_mixed_code(x, k + m, order)
    with fe.Encoder(k + m, k, 4 * S, flags=fe.CODE_MIXED_RADIX) as enc:
        if seed % 5 == 0:
            enc.set_option("fuse_radix", 0)
        what = (k, m, S, order, enc.plan())
        d = offset_view(torch, x, int(rng.integers(0, 4)))
        out = offset_view(torch, np.zeros((m, S), dtype=np.uint32), int(rng.integers(0, 4)))
        enc.encode(d, out)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy().view(np.uint32).reshape(m, S), want), what
        lost = rng.permutation(k + m)[: int(rng.integers(1, min(m, 20) + 1))]
        dp, pp = np.ones(k, np.uint8), np.ones(m, np.uint8)
        dp[lost[lost < k]] = 0
        pp[lost[lost >= k] - k] = 0
        bad_x, bad_p = x.copy(), want.copy()
        bad_x[dp == 0] = 0x11111111
        bad_p[pp == 0] = 0x22222222
        try:
            enc.decode_prepare(dp, pp)
        except fe.FastEccError as e:
            assert e.code == fe.E_UNSUPPORTED and order > (1 << 20), what  # the transform path stops at order 2^20
            return
        dd, dq = offset_view(torch, bad_x, 0), offset_view(torch, bad_p, 0)
        enc.repair(dd, dq)
        torch.cuda.synchronize()
        assert np.array_equal(dd.cpu().numpy().view(np.uint32).reshape(k, S), x), what
        assert np.array_equal(dq.cpu().numpy().view(np.uint32).reshape(m, S), want), what


@pytest.mark.parametrize("seed", range(24))
def test_random_split_decoder_configuration(torch_cuda, fe, seed):
    """The decoder's split transform (k from 2^17 up, n <= 2k on power-of-two orders): random k around 2^17 and 2^18 (powers of two and
    not), random parity counts from a few thousand to k (fold 0 ... 4, zero extension), tiny ragged blocks, random patterns from a few hundred
    losses to all the code tolerates — decode and repair against the original stripes (no oracle at these sizes: the encoder is pinned elsewhere),
    and the 2k-point form of the same pattern against both."""
    torch = torch_cuda
    rng = np.random.default_rng(7000 + seed)
    base = 1 << int(rng.integers(17, 19))
    k = int(base if rng.random() < 0.4 else rng.integers(base // 2 + 1, base + 1))
    N = 1 << (k - 1).bit_length()
    m = int(N if rng.random() < 0.3 else rng.integers(3000, N + 1))
    if k < N and m < N and rng.random() < 0.5:
        m = int(rng.integers(N // 2 + 1, N + 1))  # zero-extended data with fold 0 more often
    S = int(rng.integers(1, 24))
    g = torch.Generator(device="cuda:0").manual_seed(seed)
    data = torch.randint(0, P, (k * S,), dtype=torch.int64, device="cuda:0", generator=g).to(torch.int32)
    parity = torch.empty(m * S, dtype=torch.int32, device="cuda:0")
    count = int(rng.choice([300, 1000, m // 50 + 300, m // 3, m]))
    count = min(count, m)
    lost = rng.permutation(k + m)[:count]
    dp, pp = np.ones(k, np.uint8), np.ones(m, np.uint8)
    dp[lost[lost < k]] = 0
    pp[lost[lost >= k] - k] = 0
    with fe.Encoder(k + m, k, 4 * S) as enc:
        enc.encode(data, parity)
        for split in (1, 0):
            enc.set_option("decode_split", split)
            enc.decode_prepare(dp, pp)
            d, q = data.clone(), parity.clone()
            d.view(k, S)[torch.from_numpy(dp == 0).to("cuda:0")] = -1
            q.view(m, S)[torch.from_numpy(pp == 0).to("cuda:0")] = -2
            damaged_q = q.clone()
            enc.decode(d, q)
            torch.cuda.synchronize()
            assert torch.equal(d, data) and torch.equal(q, damaged_q), (k, m, S, count, split)
            d.view(k, S)[torch.from_numpy(dp == 0).to("cuda:0")] = -5
            enc.repair(d, q)
            torch.cuda.synchronize()
            assert torch.equal(d, data) and torch.equal(q, parity), (k, m, S, count, split)


@pytest.mark.parametrize("seed", range(24))
def test_random_p61_split_decoder_configuration(torch_cuda, fe, seed):
    """The 64-bit field's even / odd split of the (2k,k) decoder (k from 2^11 up): random k, ragged element counts, random patterns from a handful
    of lost data blocks to as many as a split form admits plus random parity losses — decode and repair against the original stripes with the
    split on and off (the encoder is pinned to the oracle elsewhere; at these sizes the two decoders check each other)."""
    torch = torch_cuda
    rng = np.random.default_rng(61700 + seed)
    k = 1 << int(rng.integers(11, 14))
    E = int(rng.choice([1, 2, 3, 5, 8]))  # 16-byte elements per block
    g = torch.Generator(device="cuda:0").manual_seed(100 + seed)
    data = torch.randint(0, P61, (k * 2 * E,), dtype=torch.int64, device="cuda:0", generator=g)
    parity = torch.empty_like(data)
    nd = int(rng.choice([1, 5, k // 64, k // 40, k // 20, k // 8]))
    npar = int(rng.choice([0, 1, 7, k // 16, k // 2, k - nd]))
    npar = min(npar, k - nd)
    dp, pp = np.ones(k, np.uint8), np.ones(k, np.uint8)
    dp[rng.permutation(k)[:nd]] = 0
    pp[rng.permutation(k)[:npar]] = 0
    dmask, pmask = torch.from_numpy(dp == 0).to("cuda:0"), torch.from_numpy(pp == 0).to("cuda:0")
    with fe.Encoder(2 * k, k, 16 * E, field=fe.FIELD_GF_P61_SQUARED) as enc:
        enc.encode(data, parity)
        for split in (1, 0):
            enc.set_option("decode_split", split)
            enc.decode_prepare(dp, pp)
            d, q = data.clone(), parity.clone()
            d.view(k, 2 * E)[dmask] = -1
            q.view(k, 2 * E)[pmask] = -2
            damaged_q = q.clone()
            enc.decode(d, q)
            torch.cuda.synchronize()
            assert torch.equal(d, data) and torch.equal(q, damaged_q), (k, E, nd, npar, split)
            d.view(k, 2 * E)[dmask] = -5
            enc.repair(d, q)
            torch.cuda.synchronize()
            assert torch.equal(d, data) and torch.equal(q, parity), (k, E, nd, npar, split)
            enc.repair(d, q)  # a second call on the same tables (the kept half-transform is rewritten)
            torch.cuda.synchronize()
            assert torch.equal(d, data) and torch.equal(q, parity), (k, E, nd, npar, split)


@pytest.mark.parametrize("seed", range(24))
def test_random_p61_coset_configuration(torch_cuda, fe, seed):
    """n = 4k / 8k over the 64-bit field: random k, element counts and plans; the codes nest (the first k parity blocks are the (2k,k) code's, pinned to
    the oracle), any n - k blocks may go."""
    from oracle import OracleP61
    orc = OracleP61()
    torch = torch_cuda
    rng = np.random.default_rng(61900 + seed)
    k = 1 << int(rng.integers(1, 12))
    e = int(rng.choice([2, 3]))
    E = int(rng.choice([1, 2, 3, 7, 16, 33]))
    if k * E > 30000:
        E = max(1, 30000 // k)
    rows = ((1 << e) - 1) * k
    x = rng.integers(0, P61, size=(k, 2 * E), dtype=np.uint64)
    x.reshape(-1)[:2] = [P61 - 1, 0]
    with fe.Encoder(k << e, k, 16 * E, field=fe.FIELD_GF_P61_SQUARED) as enc:
        plan = int(rng.choice([0, 0, 1, 2, 3, 4, 12, 13, 14, 23, 24]))
        try:
            enc.set_plan(plan)
        except fe.FastEccError:
            plan = 0
        what = (k, e, E, plan, enc.plan())
        d = torch.from_numpy(x.view(np.int64).reshape(-1).copy()).to("cuda:0")
        out = torch.empty(rows * 2 * E, dtype=torch.int64, device="cuda:0")
        enc.encode(d, out)
        torch.cuda.synchronize()
        par = out.cpu().numpy().view(np.uint64).reshape(rows, 2 * E)
        assert (par < P61).all(), what
        assert np.array_equal(par[:k], orc.encode(x)), what
        nlost = int(rng.choice([1, 2, (rows + 1) // 2, rows]))
        lost = rng.permutation(k + rows)[:nlost]
        dp, pp = np.ones(k, np.uint8), np.ones(rows, np.uint8)
        dp[lost[lost < k]] = 0
        pp[lost[lost >= k] - k] = 0
        bx, bp = x.copy(), par.copy()
        bx[dp == 0] = 3
        bp[pp == 0] = 4
        enc.decode_prepare(dp, pp)
        if seed % 3 == 0:
            enc.repair(bx, bp, mem=fe.MEM_HOST)
            assert np.array_equal(bx, x) and np.array_equal(bp, par), what
            return
        dd = torch.from_numpy(bx.view(np.int64).reshape(-1)).to("cuda:0")
        dq = torch.from_numpy(bp.view(np.int64).reshape(-1)).to("cuda:0")
        enc.decode(dd, dq)
        torch.cuda.synchronize()
        assert np.array_equal(dd.cpu().numpy().view(np.uint64).reshape(k, 2 * E), x), what
        assert np.array_equal(dq.cpu().numpy().view(np.uint64).reshape(rows, 2 * E), bp), what
        enc.repair(dd, dq)
        torch.cuda.synchronize()
        assert np.array_equal(dd.cpu().numpy().view(np.uint64).reshape(k, 2 * E), x), what
        assert np.array_equal(dq.cpu().numpy().view(np.uint64).reshape(rows, 2 * E), par), what
