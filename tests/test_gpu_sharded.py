"""GPU tests (-m gpu) of the column-sharded encode — BASELINE.json configs[3] — through the C ABI.

A GPU box for these tests has ONE device, so the slabs of fastecc_create_sharded all live on device 0
(gpu_ids = [0]*G): the partitioning, the sub-slab pipeline, both gather mechanisms and every data-placement
form run exactly as on G devices, only the "peer" copies stay inside one HBM.  Checked bit-for-bit against
the single-device HIP encode, the oracle, and at the headline size the reference's golden parity hash."""
import threading

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
    return np.random.default_rng(seed).integers(0, P, size=(N, S), dtype=np.uint64).astype(np.uint32)


@pytest.mark.parametrize("log2n,S,col0,width", [(10, 1024, 0, 1024), (10, 1024, 64, 32), (12, 256, 128, 128), (7, 96, 32, 64), (13, 40, 5, 18), (3, 8, 2, 4)])
def test_encode_columns_is_the_column_range_of_the_encode(torch_cuda, fe, oracle, log2n, S, col0, width):
    torch = torch_cuda
    N = 1 << log2n
    x = rand_stripe(log2n * 1000 + S, N, S)
    want = oracle.encode_fast(x)
    d = to_dev(torch, x)
    out = torch.full((N * S,), 0x5A5A5A5A, dtype=torch.int32, device="cuda:0")
    with fe.Encoder(2 * N, N, 4 * S) as enc:
        enc.encode_columns(d, out, col0, width)
        torch.cuda.synchronize()
    got = to_host(out).reshape(N, S)
    assert np.array_equal(got[:, col0:col0 + width], want[:, col0:col0 + width])
    untouched = np.ones(S, dtype=bool)
    untouched[col0:col0 + width] = False
    assert (got[:, untouched] == 0x5A5A5A5A).all()


def test_encode_columns_rejects_what_it_cannot_do(torch_cuda, fe):
    torch = torch_cuda
    buf = torch.zeros(64 * 64, dtype=torch.int32, device="cuda:0")
    with fe.Encoder(128, 64, 256) as enc:
        for col0, width in [(0, 0), (32, 64), (0, 65)]:
            with pytest.raises(fe.FastEccError) as ei:
                enc.encode_columns(buf, buf, col0, width)
            assert ei.value.code == fe.E_INVAL
    with fe.Encoder(64 + 16, 64, 256) as enc:  # a folded code works through scratch stripes: one piece only
        with pytest.raises(fe.FastEccError) as ei:
            enc.encode_columns(buf, buf, 0, 32)
        assert ei.value.code == fe.E_UNSUPPORTED


@pytest.mark.parametrize("G", [1, 2, 8])
@pytest.mark.parametrize("sub_slabs,gather_mode", [(1, 1), (2, 1), (2, 2), (4, 2)])
def test_sharded_slabs_equal_the_single_device_encode(torch_cuda, fe, oracle, G, sub_slabs, gather_mode):
    """data already sharded (slab g on 'GPU' g) -> parity slabs and the gathered parity stripe."""
    torch = torch_cuda
    N, S = 1 << 10, 1024
    x = rand_stripe(7 + G, N, S)
    want = oracle.encode_fast(x)
    with fe.Encoder(2 * N, N, 4 * S) as enc:  # the single-device HIP encode
        single = torch.empty(N * S, dtype=torch.int32, device="cuda:0")
        enc.encode(to_dev(torch, x), single)
        torch.cuda.synchronize()
        assert np.array_equal(to_host(single).reshape(N, S), want)
    w = S // G
    slabs = [to_dev(torch, x[:, g * w:(g + 1) * w]) for g in range(G)]
    pslabs = [torch.empty(N * w, dtype=torch.int32, device="cuda:0") for _ in range(G)]
    parity = torch.empty(N * S, dtype=torch.int32, device="cuda:0")
    with fe.ShardedEncoder(2 * N, N, 4 * S, [0] * G) as senc:
        senc.set_option("sub_slabs", sub_slabs)
        senc.set_option("gather_mode", gather_mode)
        assert "%d slabs" % G in senc.plan()
        senc.encode_sharded(slabs, pslabs, parity)          # both outputs
        torch.cuda.synchronize()
        assert np.array_equal(to_host(parity).reshape(N, S), want)
        for g in range(G):
            assert np.array_equal(to_host(pslabs[g]).reshape(N, w), want[:, g * w:(g + 1) * w])
        parity.zero_()
        senc.encode_sharded(slabs, None, parity)            # gather only, library-owned slab buffers
        torch.cuda.synchronize()
        assert np.array_equal(to_host(parity).reshape(N, S), want)
        for t in pslabs:
            t.zero_()
        senc.encode_sharded(slabs, pslabs, None)            # parity stays sharded
        torch.cuda.synchronize()
        for g in range(G):
            assert np.array_equal(to_host(pslabs[g]).reshape(N, w), want[:, g * w:(g + 1) * w])
        with pytest.raises(fe.FastEccError):
            senc.encode_sharded(slabs, None, None)


def test_pageable_host_stripes_go_through_every_slabs_staging_rings(torch_cuda, fe, oracle):
    """FASTECC_MEM_HOST on a sharded context at a size whose slabs use the rings of pinned slots (32 MiB and more per slab): a host thread per slab,
    rows gathered / scattered at a pitch, twice on one context (slots reused), and a fault injected in one slab surfaces as the call's error."""
    N, S, G = 1 << 14, 1024, 2
    x = rand_stripe(1234, N, S)
    keep = x.copy()
    want = oracle.encode_fast(x)
    with fe.ShardedEncoder(2 * N, N, 4 * S, [0] * G) as senc:
        for _ in range(2):
            out = np.full_like(x, 0x3C3C3C3C)
            senc.encode(x, out, mem=fe.MEM_HOST)
            assert np.array_equal(out, want)
        senc.set_option("inject_fault", 2)
        with pytest.raises(fe.FastEccError):
            senc.encode(x, out, mem=fe.MEM_HOST)
        out = np.full_like(x, 0x3C3C3C3C)
        senc.encode(x, out, mem=fe.MEM_HOST)  # the context is usable afterwards
        assert np.array_equal(out, want)
    assert np.array_equal(x, keep)


@pytest.mark.parametrize("mem", ["device", "host", "pinned"])
def test_sharded_context_takes_full_stripes_like_fastecc_encode(torch_cuda, fe, oracle, mem):
    torch = torch_cuda
    N, S, G = 1 << 9, 512, 4
    x = rand_stripe(99, N, S)
    want = oracle.encode_fast(x)
    with fe.ShardedEncoder(2 * N, N, 4 * S, [0] * G) as senc:
        if mem == "device":
            d = to_dev(torch, x)
            out = torch.empty_like(d)
            senc.encode(d, out, stream=torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            assert np.array_equal(to_host(out).reshape(N, S), want)
            senc.encode(d)  # in place, like the reference
            torch.cuda.synchronize()
            assert np.array_equal(to_host(d).reshape(N, S), want)
        elif mem == "host":
            out = np.empty_like(x)
            senc.encode(x, out, mem=fe.MEM_HOST)
            assert np.array_equal(out, want)
        else:
            hx = torch.from_numpy(x.view(np.int32)).pin_memory()
            hout = torch.empty_like(hx).pin_memory()
            senc.encode(hx, hout, mem=fe.MEM_HOST_PINNED)
            torch.cuda.synchronize()
            assert np.array_equal(hout.numpy().view(np.uint32), want)


def test_sharded_other_codes_and_the_64_bit_field(torch_cuda, fe, oracle):
    torch = torch_cuda
    # fewer parity blocks (a sub-coset of the (2k,k) parity) and a zero-extended code: each slab is an ordinary context
    N, S, G = 256, 256, 4
    x = rand_stripe(5, N, S)
    full = oracle.encode_fast(x)
    with fe.ShardedEncoder(N + N // 4, N, 4 * S, [0] * G) as senc:
        d, out = to_dev(torch, x), torch.empty(N // 4 * S, dtype=torch.int32, device="cuda:0")
        senc.encode(d, out)
        torch.cuda.synchronize()
        assert np.array_equal(to_host(out).reshape(N // 4, S), full[::4])
    K, Mu = 200, 50
    xz = np.zeros((N, S), dtype=np.uint32)
    xz[:K] = x[:K]
    fullz = oracle.encode_fast(xz)
    with fe.ShardedEncoder(K + Mu, K, 4 * S, [0] * G) as senc:
        d, out = to_dev(torch, x[:K]), torch.empty(Mu * S, dtype=torch.int32, device="cuda:0")
        senc.encode(d, out)
        torch.cuda.synchronize()
        assert np.array_equal(to_host(out).reshape(Mu, S), fullz[::4][:Mu])
    # GF((2^61-1)^2): 16-byte elements, slabs of whole elements
    from oracle import OracleP61
    o61 = OracleP61()
    N, elems, G = 128, 64, 4
    h = o61.fill_splitmix(N, elems, 0x77)
    want = o61.encode(h)
    d = torch.from_numpy(h.view(np.int64)).to("cuda:0")
    out = torch.empty_like(d)
    with fe.ShardedEncoder(2 * N, N, 16 * elems, [0] * G, field=fe.FIELD_GF_P61_SQUARED) as senc:
        for sub in (1, 2):  # 16 elements per slab: one or two sub-slabs of whole 128-byte row segments
            senc.set_option("sub_slabs", sub)
            out.zero_()
            senc.encode(d, out)
            torch.cuda.synchronize()
            assert np.array_equal(out.cpu().numpy().view(np.uint64).reshape(want.shape), want), sub
    # fastecc_encode_columns in this field: ranges of whole 16-byte elements (given in 4-byte words)
    with fe.Encoder(2 * N, N, 16 * elems, field=fe.FIELD_GF_P61_SQUARED) as enc:
        part = torch.full_like(d, 0x1234)
        enc.encode_columns(d, part, 4 * 8, 4 * 24)
        torch.cuda.synchronize()
        got = part.cpu().numpy().view(np.uint64).reshape(want.shape)
        assert np.array_equal(got[:, 16:64], want[:, 16:64]) and (got[:, :16] == 0x1234).all() and (got[:, 64:] == 0x1234).all()
        with pytest.raises(fe.FastEccError) as ei:
            enc.encode_columns(d, part, 2, 8)  # not whole elements
        assert ei.value.code == fe.E_INVAL


def test_sharded_argument_checks(torch_cuda, fe):
    with pytest.raises(fe.FastEccError) as ei:
        fe.ShardedEncoder(256, 128, 4096, [])
    assert ei.value.code == fe.E_INVAL
    with pytest.raises(fe.FastEccError) as ei:
        fe.ShardedEncoder(256, 128, 4096 + 4, [0] * 8)  # the block does not split into 8 slabs of whole words
    assert ei.value.code == fe.E_INVAL
    with pytest.raises(fe.FastEccError) as ei:
        fe.ShardedEncoder(256, 128, 4096, [0, 99])
    assert ei.value.code == fe.E_INVAL
    torch = torch_cuda
    buf = torch.zeros(128 * 1024, dtype=torch.int32, device="cuda:0")
    with fe.ShardedEncoder(256, 128, 4096, [0, 0]) as senc:
        for call in (lambda: senc.ntt(buf), lambda: senc.check_range(buf), lambda: senc.encode_batch(buf, buf, 1),
                     lambda: senc.encode_columns(buf, buf, 0, 32)):
            with pytest.raises(fe.FastEccError) as ei:
                call()
            assert ei.value.code == fe.E_UNSUPPORTED


def test_headline_stripe_in_eight_slabs_reproduces_the_reference_hash(torch_cuda, fe, oracle, golden_hashes):
    """(2^20, 2^19) x 4 KB split into 8 slabs of 512 B per block (configs[3]) == the reference's parity."""
    torch = torch_cuda
    N, S, G = 1 << 19, 1024, 8
    c = [g for g in golden_hashes["survey_appendix_b"] if g["input"] == "splitmix"][0]
    x = oracle.fill_splitmix(N, S, golden_hashes["splitmix_seed"])
    assert oracle.hash(x) == c["hash_input"]
    w = S // G
    slabs = [to_dev(torch, x[:, g * w:(g + 1) * w]) for g in range(G)]
    parity = torch.empty(N * S, dtype=torch.int32, device="cuda:0")
    with fe.ShardedEncoder(2 * N, N, 4 * S, [0] * G) as senc:
        for mode in (1, 2):
            parity.zero_()
            senc.set_option("gather_mode", mode)
            senc.encode_sharded(slabs, None, parity)
            torch.cuda.synchronize()
            assert oracle.hash(to_host(parity).reshape(N, S)) == c["hash_parity"], mode
        d = to_dev(torch, x)  # and the drop-in form: the same fastecc_encode call on full stripes
        senc.encode(d)
        torch.cuda.synchronize()
        assert oracle.hash(to_host(d).reshape(N, S)) == c["hash_parity"]


def test_one_context_shared_by_threads_and_streams(torch_cuda, fe, oracle):
    """fastecc.h: calls on one context are serialised and carry no state in it — two threads on two streams, a plain
    (2k,k) context (no internal buffers: device work may overlap) and a zero-extended one (internal work stripe)."""
    torch = torch_cuda
    N, S = 1 << 11, 256
    xs = [rand_stripe(40 + i, N, S) for i in range(2)]
    wants = [oracle.encode_fast(x) for x in xs]
    K, Mu = 1500, 300
    zs = []
    for x in xs:
        z = np.zeros_like(x)
        z[:K] = x[:K]
        zs.append(oracle.encode_fast(z)[::4][:Mu])
    for general in (False, True):
        enc = fe.Encoder(K + Mu, K, 4 * S) if general else fe.Encoder(2 * N, N, 4 * S)
        streams = [torch.cuda.Stream() for _ in range(2)]
        ins = [to_dev(torch, x[:K] if general else x) for x in xs]
        outs = [torch.empty((Mu if general else N) * S, dtype=torch.int32, device="cuda:0") for _ in range(2)]
        torch.cuda.synchronize()
        errors = []

        def work(i):
            try:
                for _ in range(20):
                    enc.encode(ins[i], outs[i], stream=streams[i].cuda_stream)
            except Exception as e:  # noqa: BLE001
                errors.append(e)

        threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        torch.cuda.synchronize()
        assert not errors, errors
        for i in range(2):
            got = to_host(outs[i]).reshape(-1, S)
            assert np.array_equal(got, zs[i] if general else wants[i]), (general, i)
        enc.close()


@pytest.mark.parametrize("field", ["u32", "p61"])
def test_sharded_decode_and_repair(torch_cuda, fe, oracle, field):
    """A lost block is lost in every slab: each slab repairs its own columns with the same pattern."""
    torch = torch_cuda
    N, G = 512, 4
    rng = np.random.default_rng(17)
    if field == "u32":
        S = 256
        x = rand_stripe(3, N, S)
        par = oracle.encode_fast(x)
        mk = lambda a: to_dev(torch, a)  # noqa: E731
        back = lambda t: to_host(t).reshape(N, -1)  # noqa: E731
        ctx = fe.ShardedEncoder(2 * N, N, 4 * S, [0] * G)
        bad = np.uint32(0xFFFFFFFF)
    else:
        from oracle import OracleP61
        o61 = OracleP61()
        x = o61.fill_splitmix(N, 32, 0x99)
        par = o61.encode(x)
        mk = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to("cuda:0")  # noqa: E731
        back = lambda t: t.cpu().numpy().view(np.uint64).reshape(N, -1)  # noqa: E731
        ctx = fe.ShardedEncoder(2 * N, N, 16 * 32, [0] * G, field=fe.FIELD_GF_P61_SQUARED)
        bad = np.uint64(0xFFFFFFFFFFFFFFFF)
    lost = rng.permutation(2 * N)[:N]
    dp, pp = np.ones(N, np.uint8), np.ones(N, np.uint8)
    dp[lost[lost < N]] = 0
    pp[lost[lost >= N] - N] = 0
    damaged, dpar = x.copy(), par.copy()
    damaged[dp == 0] = bad
    dpar[pp == 0] = bad
    with ctx as senc:
        with pytest.raises(fe.FastEccError):
            senc.decode(mk(damaged), mk(dpar))  # no pattern yet
        senc.decode_prepare(dp, pp)
        d, q = mk(damaged), mk(dpar)
        senc.decode(d, q)
        torch.cuda.synchronize()
        assert np.array_equal(back(d), x) and np.array_equal(back(q), dpar)
        d, q = mk(damaged), mk(dpar)
        senc.repair(d, q)
        torch.cuda.synchronize()
        assert np.array_equal(back(d), x) and np.array_equal(back(q), par)
        if field == "u32":  # host stripes: every slab travels over its GPU's own link
            hd, hq = damaged.copy(), dpar.copy()
            senc.repair(hd, hq, mem=fe.MEM_HOST)
            assert np.array_equal(hd, x) and np.array_equal(hq, par)


@pytest.mark.parametrize("fault_at", [1, 3, 6, 8])
def test_a_failure_half_way_leaves_the_context_usable(torch_cuda, fe, oracle, fault_at):
    """Fault injection (option "inject_fault" = i: the i-th slab step of the next call reports a device error after its copies and kernels
    have been enqueued).  The call must return the error only after the forked work has been waited for — the caller may free or reuse its
    buffers at once — and the following calls on the same context (encode, decode, repair) must be correct."""
    torch = torch_cuda
    N, S, G = 1 << 10, 512, 4
    x = rand_stripe(fault_at, N, S)
    want = oracle.encode_fast(x)
    with fe.ShardedEncoder(2 * N, N, 4 * S, [0] * G) as senc:
        senc.set_option("sub_slabs", 2)
        for rounds in range(2):
            d = to_dev(torch, x)
            out = torch.full_like(d, 0x11111111)
            senc.set_option("inject_fault", fault_at)
            with pytest.raises(fe.FastEccError) as ei:
                senc.encode(d, out, stream=torch.cuda.current_stream().cuda_stream)
            assert ei.value.code == fe.E_DEVICE
            # the buffers may be dropped right away: nothing of the failed call is still in flight
            del out
            d.fill_(0)
            torch.cuda.synchronize()
            d = to_dev(torch, x)
            out = torch.empty_like(d)
            senc.encode(d, out, stream=torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            assert np.array_equal(to_host(out).reshape(N, S), want), (fault_at, rounds)
        # the decoder's entry point
        dp, pp = np.ones(N, np.uint8), np.ones(N, np.uint8)
        dp[[3, 77, 500]] = 0
        pp[[9]] = 0
        damaged, dpar = x.copy(), want.copy()
        damaged[dp == 0] = 0xFFFFFFFF
        dpar[pp == 0] = 0xFFFFFFFF
        senc.decode_prepare(dp, pp)
        senc.set_option("inject_fault", min(fault_at, G))
        with pytest.raises(fe.FastEccError):
            senc.repair(to_dev(torch, damaged), to_dev(torch, dpar))
        dd, dq = to_dev(torch, damaged), to_dev(torch, dpar)
        senc.repair(dd, dq)
        torch.cuda.synchronize()
        assert np.array_equal(to_host(dd).reshape(N, S), x) and np.array_equal(to_host(dq).reshape(N, S), want)


@pytest.mark.parametrize("G", [1, 2, 4, 8])
@pytest.mark.parametrize("sub_slabs,gather_mode", [(1, 1), (2, 1), (2, 2), (4, 2)])
def test_block_distributed_all_to_all_equals_the_single_device_encode(torch_cuda, fe, oracle, G, sub_slabs, gather_mode):
    """fastecc_encode_sharded_blocks: 'GPU' g ends with parity blocks [g*M/G, (g+1)*M/G) whole; the data either in column slabs or
    block-distributed as well (the mirror transpose in front of the encode)."""
    torch = torch_cuda
    N, S = 1 << 10, 1024
    x = rand_stripe(70 + G, N, S)
    want = oracle.encode_fast(x)
    w, rows = S // G, N // G
    slabs = [to_dev(torch, x[:, g * w:(g + 1) * w]) for g in range(G)]
    blocks = [to_dev(torch, x[g * rows:(g + 1) * rows]) for g in range(G)]
    with fe.ShardedEncoder(2 * N, N, 4 * S, [0] * G) as senc:
        senc.set_option("sub_slabs", sub_slabs)
        senc.set_option("gather_mode", gather_mode)
        for data, as_blocks in ((slabs, False), (blocks, True), (slabs, False)):
            out = [torch.full((rows * S,), 0x33333333, dtype=torch.int32, device="cuda:0") for _ in range(G)]
            senc.encode_sharded_blocks(data, out, data_is_blocks=as_blocks, stream=torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            for g in range(G):
                assert np.array_equal(to_host(out[g]).reshape(rows, S), want[g * rows:(g + 1) * rows]), (g, as_blocks)
        # the gather-to-root form still works on the same context afterwards (shared slab buffers and events)
        parity = torch.empty(N * S, dtype=torch.int32, device="cuda:0")
        senc.encode_sharded(slabs, None, parity)
        torch.cuda.synchronize()
        assert np.array_equal(to_host(parity).reshape(N, S), want)


def test_block_distributed_other_codes_and_arguments(torch_cuda, fe, oracle):
    torch = torch_cuda
    # a folded code (k/4 parity blocks) and the 64-bit field: every slab is an ordinary context, only the row counts differ
    N, S, G = 1 << 9, 256, 4
    x = rand_stripe(5, N, S)
    want = oracle.encode_fast(x)[::4]
    M = N // 4
    blocks = [to_dev(torch, x[g * (N // G):(g + 1) * (N // G)]) for g in range(G)]
    out = [torch.empty((M // G) * S, dtype=torch.int32, device="cuda:0") for _ in range(G)]
    with fe.ShardedEncoder(N + M, N, 4 * S, [0] * G) as senc:
        senc.encode_sharded_blocks(blocks, out, data_is_blocks=True)
        torch.cuda.synchronize()
        for g in range(G):
            assert np.array_equal(to_host(out[g]).reshape(M // G, S), want[g * (M // G):(g + 1) * (M // G)])
    from oracle import OracleP61
    o61 = OracleP61()
    N, elems, G = 256, 64, 2
    h = o61.fill_splitmix(N, elems, 0x77)
    want61 = o61.encode(h)
    hv = h.reshape(N, 2 * elems)
    blocks = [torch.from_numpy(np.ascontiguousarray(hv[g * (N // G):(g + 1) * (N // G)]).view(np.int64)).to("cuda:0") for g in range(G)]
    out = [torch.empty((N // G) * 2 * elems, dtype=torch.int64, device="cuda:0") for _ in range(G)]
    with fe.ShardedEncoder(2 * N, N, 16 * elems, [0] * G, field=fe.FIELD_GF_P61_SQUARED) as senc:
        senc.encode_sharded_blocks(blocks, out, data_is_blocks=True)
        torch.cuda.synchronize()
        got = np.concatenate([t.cpu().numpy().view(np.uint64).reshape(N // G, 2 * elems) for t in out])
        assert np.array_equal(got, want61.reshape(N, 2 * elems))
    # blocks that do not divide over the GPUs are refused, as is an unknown layout
    buf = torch.zeros(96 * 96, dtype=torch.int32, device="cuda:0")
    with fe.ShardedEncoder(64 + 6, 64, 4 * 96, [0] * 4) as senc:  # 6 parity blocks over 4 GPUs
        with pytest.raises(fe.FastEccError) as ei:
            senc.encode_sharded_blocks([buf] * 4, [buf] * 4)
        assert ei.value.code == fe.E_INVAL
    with fe.ShardedEncoder(128, 64, 4 * 96, [0] * 3) as senc:
        with pytest.raises(fe.FastEccError) as ei:  # 64 data blocks over 3 GPUs
            senc.encode_sharded_blocks([buf] * 3, [buf] * 3, data_is_blocks=True)
        assert ei.value.code == fe.E_INVAL
        assert fe.lib().fastecc_encode_sharded_blocks(senc._h, None, 0, None, None) == fe.E_INVAL


def test_headline_stripe_block_distributed_reproduces_the_reference_hash(torch_cuda, fe, oracle, golden_hashes):
    """(2^20, 2^19) x 4 KB, data AND parity block-distributed over 8 'GPUs' (BASELINE configs[3] "block-sharded"): the eight result pieces put
    together have the reference's parity hash (2896482084)."""
    torch = torch_cuda
    N, S, G = 1 << 19, 1024, 8
    c = [g for g in golden_hashes["survey_appendix_b"] if g["input"] == "splitmix"][0]
    x = oracle.fill_splitmix(N, S, golden_hashes["splitmix_seed"])
    rows = N // G
    blocks = [to_dev(torch, x[g * rows:(g + 1) * rows]) for g in range(G)]
    out = [torch.empty(rows * S, dtype=torch.int32, device="cuda:0") for _ in range(G)]
    with fe.ShardedEncoder(2 * N, N, 4 * S, [0] * G) as senc:
        for mode in (2, 1):
            for t in out:
                t.zero_()
            senc.set_option("gather_mode", mode)
            senc.encode_sharded_blocks(blocks, out, data_is_blocks=True)
            torch.cuda.synchronize()
            got = np.concatenate([to_host(t).reshape(rows, S) for t in out])
            assert oracle.hash(got) == c["hash_parity"], mode


@pytest.mark.parametrize("fault_at", [1, 2, 5, 8])
def test_a_failure_half_way_through_the_all_to_all_leaves_the_context_usable(torch_cuda, fe, oracle, fault_at):
    torch = torch_cuda
    N, S, G = 1 << 10, 512, 4
    x = rand_stripe(40 + fault_at, N, S)
    want = oracle.encode_fast(x)
    rows = N // G
    with fe.ShardedEncoder(2 * N, N, 4 * S, [0] * G) as senc:
        senc.set_option("sub_slabs", 2)
        for mode in (2, 1):
            senc.set_option("gather_mode", mode)
            blocks = [to_dev(torch, x[g * rows:(g + 1) * rows]) for g in range(G)]
            out = [torch.full((rows * S,), 0x11111111, dtype=torch.int32, device="cuda:0") for _ in range(G)]
            senc.set_option("inject_fault", fault_at)
            with pytest.raises(fe.FastEccError) as ei:
                senc.encode_sharded_blocks(blocks, out, data_is_blocks=True)
            assert ei.value.code == fe.E_DEVICE
            del out  # nothing of the failed call is still in flight
            torch.cuda.synchronize()
            out = [torch.empty(rows * S, dtype=torch.int32, device="cuda:0") for _ in range(G)]
            senc.encode_sharded_blocks(blocks, out, data_is_blocks=True)
            torch.cuda.synchronize()
            for g in range(G):
                assert np.array_equal(to_host(out[g]).reshape(rows, S), want[g * rows:(g + 1) * rows]), (mode, g)


def test_sharded_decode_at_2_17_blocks_takes_the_split_transform(torch_cuda, fe):
    """k = 2^17 in four column slabs on one device: every slab's decoder is set up on its own host thread (the same pattern), decodes through
    the split transform (two half-size transforms, tests/test_gpu_decode.py) and repairs; compared on the device with the original stripes."""
    torch = torch_cuda
    N, S, G = 1 << 17, 64, 4
    g = torch.Generator(device="cuda:0").manual_seed(99)
    data = torch.randint(0, P, (N * S,), dtype=torch.int64, device="cuda:0", generator=g).to(torch.int32)
    parity = torch.empty_like(data)
    rng = np.random.default_rng(17)
    with fe.ShardedEncoder(2 * N, N, 4 * S, [0] * G) as senc:
        senc.encode(data, parity)
        torch.cuda.synchronize()
        for count in (N // 3, 500):
            lost = rng.permutation(2 * N)[:count]
            dp, pp = np.ones(N, np.uint8), np.ones(N, np.uint8)
            dp[lost[lost < N]] = 0
            pp[lost[lost >= N] - N] = 0
            senc.decode_prepare(dp, pp)
            damaged, dpar = data.clone(), parity.clone()
            damaged.view(N, S)[torch.from_numpy(dp == 0).to("cuda:0")] = -1
            dpar.view(N, S)[torch.from_numpy(pp == 0).to("cuda:0")] = -2
            senc.decode(damaged, dpar)
            torch.cuda.synchronize()
            assert bool((damaged == data).all())
            damaged.view(N, S)[torch.from_numpy(dp == 0).to("cuda:0")] = -3
            senc.repair(damaged, dpar)
            torch.cuda.synchronize()
            assert bool((damaged == data).all()) and bool((dpar == parity).all())


def test_sharded_in_place_needs_room_for_the_parity(torch_cuda, fe):
    """parity == data with n - k > k would write past the data stripe: rejected like on one device."""
    torch = torch_cuda
    k, S, G = 64, 64, 2
    with fe.ShardedEncoder(4 * k, k, 4 * S, [0] * G) as senc:
        d = torch.zeros(k * S, dtype=torch.int32, device="cuda:0")
        with pytest.raises(fe.FastEccError) as ei:
            senc.encode(d)  # in place
        assert ei.value.code == fe.E_INVAL
