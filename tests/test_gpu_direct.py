"""GPU tests (-m gpu) of direct.hip: lost (or parity) blocks as fixed linear combinations of other blocks, by both kernels —
the VALU form (96-bit lazy accumulation, sweeps of 16 outputs) and the MFMA form (signed base-256 digits on
v_mfma_i32_32x32x32_i8) — against each other, the transform path, the original stripes and the oracle.

The reference documents the decoder (README.md:83-119, RS.md:42-79) and implements none of it; its encoder (RS.cpp:40-63) evaluates
the same polynomial these kernels evaluate from the Lagrange basis, so the parity checks are pinned through the oracle.  Bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

P = 0xFFF00001
KERNELS = {"valu": 1, "mfma": 2}


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


def lose(rng, k, m, e, data_only=False):
    """e lost blocks, at least one of them a data block -> (data_present, parity_present)"""
    if data_only:
        lost = rng.permutation(k)[:e]
    else:
        lost = np.unique(np.r_[int(rng.integers(0, k)), rng.permutation(k + m)[: e - 1]])
    dp, pp = np.ones(k, np.uint8), np.ones(m, np.uint8)
    dp[lost[lost < k]] = 0
    pp[lost[lost >= k] - k] = 0
    return dp, pp


def edge_values(rng, shape):
    """random words with the values around the digit trick's switch-over and the ends of the field sprinkled in"""
    x = rng.integers(0, P, size=shape, dtype=np.uint64).astype(np.uint32)
    special = np.array([0, 1, P - 1, P - 2, 0x7F7F7F7F, 0x7F7F7F80, 0x7F7F7F81, 0x80808080, 0x807F7F7F, 0x80000000, 0x7FFFFFFF, 0xFFEFFFFF], dtype=np.uint32)
    idx = rng.integers(0, x.size, size=max(8, x.size // 50))
    x.reshape(-1)[idx] = special[rng.integers(0, len(special), size=len(idx))]
    return x


@pytest.mark.parametrize("N,S", [(1 << 12, 64), (1 << 13, 130), (1 << 11, 256), (1 << 12, 33), (64, 64), (32, 66)])
def test_both_kernels_recover_up_to_256_lost_blocks(torch_cuda, fe, oracle, N, S):
    """1 ... 256 lost blocks of a (2k,k) codeword: VALU kernel == MFMA kernel == transform path == the original stripes (decode and
    repair).  Odd row lengths and short stripes fall back to the VALU kernel whatever is asked for."""
    torch = torch_cuda
    rng = np.random.default_rng(N + S)
    x = edge_values(rng, (N, S))
    par = oracle.encode_fast(x)
    with fe.Encoder(2 * N, N, 4 * S) as enc:
        for e in (1, 2, 5, 8, 15, 16, 17, 24, 32, 33, 48, 64, 65, 100, 128, 129, 200, 256):
            if e > N:
                continue
            dp, pp = lose(rng, N, N, e, data_only=(e % 3 == 0))
            bad_x, bad_p = x.copy(), par.copy()
            bad_x[dp == 0] = 0xA5A5A5A5
            bad_p[pp == 0] = 0x5A5A5A5A
            got = {}
            for name, direct_max, kernel in (("valu", 256, 1), ("mfma", 256, 2), ("transform", 0, 0)):
                if name == "transform" and e not in (1, 17, 100):
                    continue
                enc.set_option("decode_direct_max", direct_max)
                enc.set_option("direct_kernel", kernel)
                enc.decode_prepare(dp, pp)
                d, q = to_dev(torch, bad_x), to_dev(torch, bad_p)
                enc.decode(d, q)
                assert (to_host(d, (N, S)) == x).all(), (e, name)
                assert (to_host(q, (N, S)) == bad_p).all(), (e, name)  # decode leaves the parity alone
                enc.repair(d, q)
                assert (to_host(d, (N, S)) == x).all() and (to_host(q, (N, S)) == par).all(), (e, name)
                got[name] = to_host(d, (N, S)).copy()
            assert (got["valu"] == got["mfma"]).all()
        enc.set_option("direct_kernel", 0)
        enc.set_option("decode_direct_max", 256)
        with pytest.raises(fe.FastEccError):
            enc.set_option("decode_direct_max", 257)
        with pytest.raises(fe.FastEccError):
            enc.set_option("direct_kernel", 3)


@pytest.mark.parametrize("k,S", [(1 << 12, 64), (5000, 96), (3 << 10, 66)])
def test_codes_with_up_to_256_parity_blocks_are_encoded_directly(torch_cuda, fe, oracle, k, S):
    """n - k = 1 ... 256: parity straight from the Lagrange basis by either kernel == the transform pipeline (encode_direct_max = 0);
    for the power-of-two k also == the oracle (the first n - k blocks of the (2k,k) parity... of the sub-coset layout are the
    pipeline's own job: here the pipeline is the pinned side)."""
    torch = torch_cuda
    rng = np.random.default_rng(k + S)
    x = edge_values(rng, (k, S))
    for m in (1, 7, 16, 17, 32, 40, 64, 100, 128, 200, 256):
        with fe.Encoder(k + m, k, 4 * S) as enc:
            outs = {}
            for name, direct_max, kernel in (("pipeline", 0, 0), ("valu", 256, 1), ("mfma", 256, 2), ("auto", 256, 0)):
                enc.set_option("encode_direct_max", direct_max)
                enc.set_option("direct_kernel", kernel)
                out = torch.full((m * S,), 0x66666666, dtype=torch.int32, device="cuda:0")
                dx = to_dev(torch, x)
                enc.encode(dx, out)
                outs[name] = to_host(out, (m, S)).copy()
                assert (to_host(dx, (k, S)) == x).all()
            for name in ("valu", "mfma", "auto"):
                assert (outs[name] == outs["pipeline"]).all(), (k, m, name)


def test_mixed_radix_and_zero_extended_codes_by_the_mfma_kernel(torch_cuda, fe, oracle):
    """The direct path does not care what N is: a zero-extended code (k = 5000 of 8192 points) and a mixed-radix one (3 * 2^11)."""
    torch = torch_cuda
    rng = np.random.default_rng(5)
    S = 64
    for k, m, flags in ((5000, 3000, 0), (3 << 11, 3 << 11, "mixed")):
        x = edge_values(rng, (k, S))
        kw = dict(flags=fe.CODE_MIXED_RADIX) if flags == "mixed" else {}
        with fe.Encoder(k + m, k, 4 * S, **kw) as enc:
            par_dev = torch.empty(m * S, dtype=torch.int32, device="cuda:0")
            enc.encode(to_dev(torch, x), par_dev)
            par = to_host(par_dev, (m, S)).copy()
            for e in (20, 70):
                dp, pp = lose(rng, k, m, e)
                bad_x, bad_p = x.copy(), par.copy()
                bad_x[dp == 0] = 1
                bad_p[pp == 0] = 2
                for kernel in (1, 2):
                    enc.set_option("direct_kernel", kernel)
                    enc.decode_prepare(dp, pp)
                    d, q = to_dev(torch, bad_x), to_dev(torch, bad_p)
                    enc.repair(d, q)
                    assert (to_host(d, (k, S)) == x).all() and (to_host(q, (m, S)) == par).all(), (k, e, kernel)


@pytest.mark.parametrize("lost", [16, 64, 128])
def test_headline_size(torch_cuda, fe, lost):
    """(2^20, 2^19) x 4 KB: `lost` blocks gone, both kernels give back the original data and parity."""
    torch = torch_cuda
    N, S = 1 << 19, 1024
    g = torch.Generator(device="cuda:0").manual_seed(lost)
    x = (torch.randint(0, P, (N * S,), generator=g, device="cuda:0", dtype=torch.int64)).to(torch.int32)
    with fe.Encoder(2 * N, N, 4 * S) as enc:
        par = torch.empty_like(x)
        enc.encode(x, par)
        rng = np.random.default_rng(lost)
        dp, pp = lose(rng, N, N, lost)
        for kernel in (1, 2):
            enc.set_option("direct_kernel", kernel)
            enc.set_option("decode_direct_max", 256)
            enc.decode_prepare(dp, pp)
            d, q = x.clone(), par.clone()
            d.view(N, S)[torch.from_numpy(np.flatnonzero(dp == 0)).to("cuda:0")] = 0x12345678
            q.view(N, S)[torch.from_numpy(np.flatnonzero(pp == 0)).to("cuda:0")] = 0x0BADF00D
            enc.repair(d, q)
            torch.cuda.synchronize()
            assert torch.equal(d, x) and torch.equal(q, par), (lost, kernel)


@pytest.mark.parametrize("N,S", [(64, 16), (4096, 33), (1 << 15, 64)])
def test_a_repair_of_few_losses_reads_the_stripe_once(torch_cuda, fe, oracle, N, S):
    """Data AND parity lost, at most 32 blocks in all: fastecc_repair is ONE pass over the survivors (the lost parity blocks are further outputs of the
    interpolation on the data pass's nodes) — the profile counts the passes; fastecc_decode on the same pattern is that pass too and leaves the parity
    stripe alone.  Only parity lost, or more than 32 blocks: as before (one pass for the data from the survivors, one for the parity from the data)."""
    torch = torch_cuda
    rng = np.random.default_rng(N + S)
    x = rng.integers(0, P, size=(N, S), dtype=np.uint64).astype(np.uint32)
    par = oracle.encode(x)
    with fe.Encoder(2 * N, N, 4 * S) as enc:
        enc.set_option("decode_direct_max", 256)
        for nd, npar, passes_repair, passes_decode in ((1, 1, 1, 1), (8, 8, 1, 1), (3, 29, 1, 1), (31, 1, 1, 1), (0, 5, 1, 0), (5, 0, 1, 1), (17, 16, 2, 1), (2, 40, 2, 1)):
            if nd + npar > N:
                continue
            dp, pp = np.ones(N, np.uint8), np.ones(N, np.uint8)
            dp[rng.permutation(N)[:nd]] = 0
            pp[rng.permutation(N)[:npar]] = 0
            bad_x, bad_p = x.copy(), par.copy()
            bad_x[dp == 0] = 0x12345678
            bad_p[pp == 0] = 0x0BADF00D
            for kernel in (1, 2):
                enc.set_option("direct_kernel", kernel)
                enc.decode_prepare(dp, pp)
                for call, passes in (("decode", passes_decode), ("repair", passes_repair)):
                    d, q = torch.from_numpy(bad_x.view(np.int32)).to("cuda:0"), torch.from_numpy(bad_p.view(np.int32)).to("cuda:0")
                    enc.profile(True)
                    enc.profile_reset()
                    getattr(enc, call)(d, q)
                    torch.cuda.synchronize()
                    prof = enc.profile_read()
                    enc.profile(False)
                    assert prof.get("direct_pass", (0, 0, 0))[1] == passes, (nd, npar, kernel, call, prof)
                    assert (d.cpu().numpy().view(np.uint32).reshape(N, S) == x).all(), (nd, npar, kernel, call)
                    want_p = par if call == "repair" else bad_p
                    assert (q.cpu().numpy().view(np.uint32).reshape(N, S) == want_p).all(), (nd, npar, kernel, call)
        enc.set_option("direct_kernel", 0)
