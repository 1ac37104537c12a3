"""GPU tests (-m gpu) of the GF.md:72-104 data packing (fastecc_pack_blocks / fastecc_unpack_blocks).

Upstream has prose only, so the format is ours (include/fastecc.h) and parity is unpinned; the checker is the
sequential restatement in oracle/fastecc_oracle.c, itself pinned to tests/golden/golden_pack.json (independent
pure-Python statement).  Bit-exact everywhere."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

P = 0xFFF00001
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


def to_dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to("cuda:0")


def to_host(t, shape):
    return t.cpu().numpy().view(np.uint32).reshape(shape)


def gpu_pack(torch, enc, raw):
    k, W = raw.shape
    out = torch.empty(k * (W + 1), dtype=torch.int32, device="cuda:0")
    enc.pack_blocks(to_dev(torch, raw), out)
    return to_host(out, (k, W + 1))


def gpu_unpack(torch, enc, packed):
    k, S = packed.shape
    out = torch.empty(k * (S - 1), dtype=torch.int32, device="cuda:0")
    bad = enc.unpack_blocks(to_dev(torch, packed), out)
    return to_host(out, (k, S - 1)), bad


def raw_stripe(rng, k, W, fff_rate):
    x = rng.integers(0, 1 << 32, size=(k, W), dtype=np.uint64).astype(np.uint32)
    hit = rng.random((k, W)) < fff_rate
    return np.where(hit, x | np.uint32(0xFFF00000), x)


def test_golden_cases(torch_cuda, fe):
    doc = json.load(open(os.path.join(HERE, "golden", "golden_pack.json")))
    for case in doc["cases"]:
        raw = np.array(case["raw"], dtype=np.uint32)
        want = np.array(case["packed"], dtype=np.uint32)
        W = raw.size
        stripe = np.stack([raw, raw[::-1].copy()])  # k = 2; the second block only keeps the context legal
        with fe.Encoder(4, 2, 4 * (W + 1)) as enc:
            got = gpu_pack(torch_cuda, enc, stripe)
            assert (got[0] == want).all(), case["name"]
            back, bad = gpu_unpack(torch_cuda, enc, got)
            assert bad == 0 and (back == stripe).all(), case["name"]


@pytest.mark.parametrize("W", [1, 2, 63, 64, 65, 513, 1000, 1024])
@pytest.mark.parametrize("rate", [0.0, 1 / 4096, 0.02, 0.5, 1.0])
def test_pack_matches_oracle_and_round_trips(torch_cuda, fe, oracle, W, rate):
    k = 64
    raw = raw_stripe(np.random.default_rng(W * 7 + int(rate * 1000)), k, W, rate)
    want = oracle.pack_blocks(raw)
    with fe.Encoder(2 * k, k, 4 * (W + 1)) as enc:
        got = gpu_pack(torch_cuda, enc, raw)
        assert (got == want).all()
        assert (got < P).all() and (got[:, :W] < 0xFFF00000).all()
        d = to_dev(torch_cuda, got)
        assert enc.check_range(d) == 0
        back, bad = gpu_unpack(torch_cuda, enc, got)
        assert bad == 0 and (back == raw).all()


def test_unpack_rejects_what_no_packer_writes(torch_cuda, fe, oracle):
    k, W = 32, 1024
    rng = np.random.default_rng(3)
    raw = raw_stripe(rng, k, W, 0.01)
    packed = oracle.pack_blocks(raw)
    flagged = [i for i in range(k) if packed[i, W] == 1]
    plain = [i for i in range(k) if packed[i, W] == 0]
    assert len(flagged) >= 8
    bad_blocks = packed.copy()
    bad_blocks[flagged[0], W] = 2                                  # flag word out of range
    bad_blocks[flagged[1], 0] |= np.uint32(0x800 << 20)            # index entry with bit 11 set
    bad_blocks[flagged[2], 0] = (bad_blocks[flagged[2], 0] & 0xFFFFF) | np.uint32((0x400 | 1000) << 20)
    bad_blocks[flagged[2], 1] = (bad_blocks[flagged[2], 1] & 0xFFFFF) | np.uint32(999 << 20)   # not increasing
    bad_blocks[flagged[3], 1023] |= np.uint32(0xFFF00000)          # a 0xFFF digit among the values
    bad_blocks[flagged[4], :W] |= np.uint32(0x400 << 20)           # entries never end
    if plain:
        bad_blocks[plain[0], 5] |= np.uint32(0xFFF00000)           # flag 0 but a 0xFFF digit
    want, want_bad = oracle.unpack_blocks(bad_blocks)
    assert want_bad == 5 + (1 if plain else 0)
    with fe.Encoder(2 * k, k, 4 * (W + 1)) as enc:
        got, bad = gpu_unpack(torch_cuda, enc, bad_blocks)
        assert bad == want_bad
        assert (got == want).all()  # good blocks decoded, bad ones passed through unchanged
    # shorter blocks: a position >= W is out of range
    W2 = 100
    raw2 = raw_stripe(rng, 4, W2, 0.05)
    p2 = oracle.pack_blocks(raw2)
    i = int(np.argmax(p2[:, W2]))
    assert p2[i, W2] == 1
    p2[i, 0] = (p2[i, 0] & 0xFFFFF) | np.uint32(100 << 20)
    want2, wb2 = oracle.unpack_blocks(p2)
    with fe.Encoder(8, 4, 4 * (W2 + 1)) as enc:
        got2, bad2 = gpu_unpack(torch_cuda, enc, p2)
        assert bad2 == wb2 == 1 and (got2 == want2).all()


def test_sector_pipeline_4096_to_4100(torch_cuda, fe, oracle):
    """README.md:160-163: 4096-byte sectors -> 4100-byte blocks -> 4100-byte parity (bit-exact vs the oracle)."""
    torch = torch_cuda
    k, W = 1 << 10, 1024
    raw = raw_stripe(np.random.default_rng(12), k, W, 1 / 4096)
    raw[0, :] = 0xFFFFFFFF
    packed_want = oracle.pack_blocks(raw)
    parity_want = oracle.encode_fast(packed_want)
    with fe.Encoder(2 * k, k, 4100) as enc:
        d_raw = to_dev(torch, raw)
        d_packed = torch.empty(k * 1025, dtype=torch.int32, device="cuda:0")
        d_parity = torch.empty_like(d_packed)
        enc.pack_blocks(d_raw, d_packed)
        enc.encode(d_packed, d_parity)
        assert (to_host(d_parity, (k, 1025)) == parity_want).all()
        # host-memory form of both calls
        packed_host = np.empty((k, 1025), dtype=np.uint32)
        enc.pack_blocks(raw, packed_host, mem=fe.MEM_HOST)
        assert (packed_host == packed_want).all()
        raw_back = np.empty_like(raw)
        assert enc.unpack_blocks(packed_host, raw_back, mem=fe.MEM_HOST) == 0
        assert (raw_back == raw).all()


def test_padded_row_pitch(torch_cuda, fe, oracle):
    """pack -> encode -> unpack on device stripes with 4224-byte rows (row_pitch_words = 1056)."""
    torch = torch_cuda
    k, W, pitch = 1 << 11, 1024, 1056
    raw = raw_stripe(np.random.default_rng(21), k, W, 1 / 2048)
    packed_want = oracle.pack_blocks(raw)
    parity_want = oracle.encode_fast(packed_want)
    with fe.Encoder(2 * k, k, 4100) as enc:
        enc.set_option("row_pitch_words", pitch)
        d_packed = torch.full((k * pitch,), -1, dtype=torch.int32, device="cuda:0")
        d_parity = torch.full((k * pitch,), -1, dtype=torch.int32, device="cuda:0")
        enc.pack_blocks(to_dev(torch, raw), d_packed)
        got = to_host(d_packed, (k, pitch))
        assert (got[:, :W + 1] == packed_want).all() and (got[:, W + 1:] == 0xFFFFFFFF).all()  # padding untouched
        enc.encode(d_packed, d_parity)
        assert (to_host(d_parity, (k, pitch))[:, :W + 1] == parity_want).all()
        back = torch.empty(k * W, dtype=torch.int32, device="cuda:0")
        assert enc.unpack_blocks(d_packed, back) == 0
        assert (to_host(back, (k, W)) == raw).all()


def test_headline_size_round_trip(torch_cuda, fe):
    """k = 2^19 sectors of 4096 bytes (2 GiB): pack, range check, unpack; compared on the device."""
    torch = torch_cuda
    k, W = 1 << 19, 1024
    g = torch.Generator(device="cuda:0").manual_seed(5)
    raw = torch.randint(-(1 << 31), 1 << 31, (k * W,), dtype=torch.int64, device="cuda:0", generator=g).to(torch.int32)
    with fe.Encoder(2 * k, k, 4100) as enc:
        packed = torch.empty(k * (W + 1), dtype=torch.int32, device="cuda:0")
        enc.pack_blocks(raw, packed)
        assert enc.check_range(packed) == 0
        flags = packed.view(k, W + 1)[:, W]
        frac = float((flags == 1).float().mean())
        assert 0.15 < frac < 0.30   # 1 - (1 - 2^-12)^1024 = 22 % of random sectors need recoding
        back = torch.empty_like(raw)
        assert enc.unpack_blocks(packed, back) == 0
        assert bool((back == raw).all())


def test_unsupported_shapes(torch_cuda, fe):
    torch = torch_cuda
    with fe.Encoder(8, 4, 8192) as enc:  # 2047 raw words: positions do not fit 10 bits
        a = torch.zeros(4 * 2048, dtype=torch.int32, device="cuda:0")
        with pytest.raises(fe.FastEccError) as ei:
            enc.pack_blocks(a, a)
        assert ei.value.code == fe.E_UNSUPPORTED
    with fe.Encoder(8, 4, 64, field=fe.FIELD_GF_P61_SQUARED) as enc:
        a = torch.zeros(64, dtype=torch.int64, device="cuda:0")
        with pytest.raises(fe.FastEccError) as ei:
            enc.pack_blocks(a, a)
        assert ei.value.code == fe.E_UNSUPPORTED
