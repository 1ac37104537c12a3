"""GPU parity tests (-m gpu) for fewer parity than data blocks: (n,k) = (k + k/2^d, k), d = 1..4.

RS.md:13-33 ("output some M values from NTT result"): the parity of the (k + M, k) code is taken from the same
polynomial; here parity block j is block j * k/M of the reference's (2k,k) parity, so the checker is the PINNED
oracle / the unmodified reference itself, subsampled.  Bit-exact."""
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


def stripe(seed, N, S):
    return np.random.default_rng(seed).integers(0, P, size=(N, S), dtype=np.uint64).astype(np.uint32)


@pytest.mark.parametrize("logn", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13])
@pytest.mark.parametrize("d", [1, 2, 3, 4])
def test_folded_encode_is_the_subsampled_reference_parity(torch_cuda, fe, oracle, logn, d):
    if d > logn:
        pytest.skip("n - k >= 1")
    N, S = 1 << logn, 96 if logn > 10 else 200
    M = N >> d
    x = stripe(100 * logn + d, N, S)
    want = oracle.encode_fast(x)[:: 1 << d]
    with fe.Encoder(N + M, N, 4 * S) as enc:
        dx = to_dev(torch_cuda, x)
        out = torch_cuda.empty(M * S, dtype=torch_cuda.int32, device="cuda:0")
        enc.encode(dx, out)
        assert (to_host(out, (M, S)) == want).all(), enc.plan()
        assert (to_host(dx, (N, S)) == x).all()  # data untouched
        enc.encode(dx)  # in place: the first M blocks of the data buffer receive the parity
        assert (to_host(dx, (N, S))[:M] == want).all()


@pytest.mark.parametrize("plan", [11, 21, 31, 52, 54, 1060, 1070, 1080, 1090, 1100, 1101, 2100, 3090, 3100])
@pytest.mark.parametrize("d", [1, 3, 4])
def test_every_plan(torch_cuda, fe, oracle, plan, d):
    N, S = 1 << 12, 64
    M = N >> d
    x = stripe(plan + d, N, S)
    want = oracle.encode_fast(x)[:: 1 << d]
    with fe.Encoder(N + M, N, 4 * S) as enc:
        enc.set_plan(plan)
        out = torch_cuda.empty(M * S, dtype=torch_cuda.int32, device="cuda:0")
        enc.encode(to_dev(torch_cuda, x), out)
        assert (to_host(out, (M, S)) == want).all(), enc.plan()


def test_against_the_unmodified_reference(torch_cuda, fe):
    from oracle import Reference
    if not Reference.available():
        pytest.skip("oracle/_ref not built")
    ref = Reference()
    N, S, d = 1 << 11, 513, 2
    x = stripe(5, N, S)
    want = ref.encode(x)[::4]
    with fe.Encoder(N + N // 4, N, 4 * S) as enc:
        out = torch_cuda.empty((N // 4) * S, dtype=torch_cuda.int32, device="cuda:0")
        enc.encode(to_dev(torch_cuda, x), out)
        assert (to_host(out, (N // 4, S)) == want).all()


def test_host_forms_and_row_pitch(torch_cuda, fe, oracle):
    N, S, d = 256, 1025, 2
    M = N >> d
    x = stripe(9, N, S)
    want = oracle.encode_fast(x)[::4]
    with fe.Encoder(N + M, N, 4 * S) as enc:
        out = np.empty((M, S), dtype=np.uint32)
        enc.encode_host(x, out)
        assert (out == want).all()
        blocks = [np.ascontiguousarray(x[i]).copy() for i in range(N)]
        enc.encode_blocks([b.ctypes.data for b in blocks])
        assert (np.stack(blocks[:M]) == want).all()
        assert (np.stack(blocks[M:]) == x[M:]).all()  # only the first n - k blocks are overwritten
        pitch = 1056
        enc.set_option("row_pitch_words", pitch)
        padded = np.zeros((N, pitch), dtype=np.uint32)
        padded[:, :S] = x
        dout = torch_cuda.zeros(M * pitch, dtype=torch_cuda.int32, device="cuda:0")
        enc.encode(to_dev(torch_cuda, padded), dout)
        assert (to_host(dout, (M, pitch))[:, :S] == want).all()


def test_headline_k_with_quarter_parity(torch_cuda, fe):
    """k = 2^19, 4 KB blocks, n - k = 2^17: equal to every 4th block of the (2^20, 2^19) parity computed on the device
    (that parity is itself pinned by the Appendix-B hashes in test_gpu_parity.py)."""
    torch = torch_cuda
    N, S, d = 1 << 19, 1024, 2
    g = torch.Generator(device="cuda:0").manual_seed(3)
    data = torch.randint(0, P, (N * S,), dtype=torch.int64, device="cuda:0", generator=g).to(torch.int32)
    full = torch.empty_like(data)
    with fe.Encoder(2 * N, N, 4 * S) as enc:
        enc.encode(data, full)
    part = torch.empty((N >> d) * S, dtype=torch.int32, device="cuda:0")
    with fe.Encoder(N + (N >> d), N, 4 * S) as enc:
        enc.encode(data, part)
        torch.cuda.synchronize()
    assert bool((part.view(N >> d, S) == full.view(N, S)[:: 1 << d]).all())


def test_rejected_shapes(fe):
    with pytest.raises(fe.FastEccError) as ei:
        fe.Encoder(64 + 128, 64, 64)         # more parity than data blocks, and neither 4k nor 8k
    assert ei.value.code == fe.E_UNSUPPORTED
    with pytest.raises(fe.FastEccError) as ei:
        fe.Encoder(64, 64, 64)               # no parity at all
    assert ei.value.code == fe.E_INVAL
