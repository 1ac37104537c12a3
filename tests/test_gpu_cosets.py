"""GPU parity tests (-m gpu) for more parity than data blocks: n = 4k and n = 8k.

RS.md:13-33: the parity points are the n-th roots of unity that are not data points; they form 2^e - 1 cosets of
the data points, evaluated one coset at a time.  Coset 0 is the reference's own (2k,k) parity (pinned); the others
use the same composition with another coset generator, checked against the oracle's transforms
(iNTT, block i *= g^i / N, NTT — RS.cpp:40-63 with g in place of root(2N)) and, in tests/test_oracle.py, against
direct polynomial evaluation.  Bit-exact."""
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


def coset_generators(oracle, N, e):
    """w_2N; w_4N, w_4N^3; w_8N, w_8N^3, w_8N^5, w_8N^7 — the nesting order of include/fastecc.h."""
    gens = []
    for j in range(1, e + 1):
        w = oracle.gf_root(N << j)
        gens += [oracle.gf_pow(w, c) for c in range(1, 1 << j, 2)]
    return gens


def oracle_parity(oracle, x, e):
    N = x.shape[0]
    coef = oracle.ntt_fast(x, inverse=True)
    inv_n = oracle.gf_inv(N)
    return np.concatenate([oracle.ntt_fast(oracle.scale_blocks(coef, inv_n, g)) for g in coset_generators(oracle, N, e)])


@pytest.mark.parametrize("logn", [1, 2, 5, 6, 9, 10, 11, 13])
@pytest.mark.parametrize("e", [2, 3])
def test_multi_coset_parity(torch_cuda, fe, oracle, logn, e):
    N, S = 1 << logn, 77
    x = np.random.default_rng(logn * 10 + e).integers(0, P, size=(N, S), dtype=np.uint64).astype(np.uint32)
    want = oracle_parity(oracle, x, e)
    rows = ((1 << e) - 1) * N
    with fe.Encoder(N << e, N, 4 * S) as enc:
        dx = to_dev(torch_cuda, x)
        out = torch_cuda.empty(rows * S, dtype=torch_cuda.int32, device="cuda:0")
        enc.encode(dx, out)
        got = to_host(out, (rows, S))
        assert (got[:N] == oracle.encode_fast(x)).all()   # codes nest: the first k parity blocks are the (2k,k) parity
        assert (got == want).all(), enc.plan()
        assert (to_host(dx, (N, S)) == x).all()
        host_out = np.empty((rows, S), dtype=np.uint32)
        enc.encode_host(x, host_out)
        assert (host_out == want).all()
        with pytest.raises(fe.FastEccError):
            enc.encode(dx)  # in place is impossible: the parity is larger than the data


@pytest.mark.parametrize("plan", [21, 52, 1080, 2100, 3100])
def test_plans(torch_cuda, fe, oracle, plan):
    N, S, e = 1 << 12, 40, 2
    x = np.random.default_rng(plan).integers(0, P, size=(N, S), dtype=np.uint64).astype(np.uint32)
    want = oracle_parity(oracle, x, e)
    with fe.Encoder(N << e, N, 4 * S) as enc:
        enc.set_plan(plan)
        out = torch_cuda.empty(3 * N * S, dtype=torch_cuda.int32, device="cuda:0")
        enc.encode(to_dev(torch_cuda, x), out)
        assert (to_host(out, (3 * N, S)) == want).all(), enc.plan()


def test_limits(fe):
    with pytest.raises(fe.FastEccError) as ei:
        fe.Encoder(1 << 21, 1 << 19, 64)   # w_(2^21) does not exist in GF(0xFFF00001)
    assert ei.value.code == fe.E_UNSUPPORTED
    with pytest.raises(fe.FastEccError) as ei:
        fe.Encoder(16 * 64, 64, 64)
    assert ei.value.code == fe.E_UNSUPPORTED
