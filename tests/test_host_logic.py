"""CPU tests of the host-side planning logic behind the C ABI (no device needed): plan selection and the
level-packed twiddle tables, checked entry by entry against the oracle's field arithmetic."""
import ctypes

import numpy as np
import pytest

P = 0xFFF00001


def describe(lib, k, block_bytes, plan=0):
    buf = ctypes.create_string_buffer(512)
    lib.fastecc_plan_describe.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t]
    rc = lib.fastecc_plan_describe(k, block_bytes, plan, buf, 512)
    return rc, buf.value.decode()


def twiddles(lib, k, block_bytes, plan, which):
    lib.fastecc_plan_twiddles.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int, ctypes.c_int,
                                          ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_int32)]
    n = k.bit_length() - 1
    out = (ctypes.c_uint32 * k)()
    sl = (ctypes.c_int32 * n)()
    rc = lib.fastecc_plan_twiddles(k, block_bytes, plan, which, out, sl)
    assert rc == 0
    return np.frombuffer(out, dtype=np.uint32).copy(), list(sl)


def parse(text):
    """'T32:dif9@10,T32:mid10@0,T32:dit9@10 v1' -> [(kind, mode, levels, s), ...]"""
    passes = []
    for item in text.split(" ")[0].split(","):
        kind = "reg"
        if ":" in item:
            kind, item = item.split(":")
        mode, rest = item[:3], item[3:]
        levels, s = rest.split("@")
        passes.append((kind, mode, int(levels), int(s)))
    return passes


def test_headline_plan_is_three_trips(hip_lib):
    rc, text = describe(hip_lib, 1 << 19, 4096)
    assert rc == 0
    # the default gives MID one level fewer than it could take (VALU-bound) and the HBM-bound outer passes one more
    assert parse(text) == [("T32", "dif", 10, 9), ("T32", "mid", 9, 0), ("T32", "dit", 10, 9)], text
    assert parse(describe(hip_lib, 1 << 18, 4096)[1]) == [("T64", "dif", 9, 9), ("T32", "mid", 9, 0), ("T64", "dit", 9, 9)]
    assert parse(describe(hip_lib, 1 << 17, 4096)[1]) == [("S32", "dif", 8, 9), ("T32", "mid", 9, 0), ("S32", "dit", 8, 9)]
    assert parse(describe(hip_lib, 1 << 16, 4096)[1]) == [("S32", "dif", 8, 8), ("T32", "mid", 8, 0), ("S32", "dit", 8, 8)]
    assert parse(describe(hip_lib, 1 << 15, 4096)[1])[1] == ("T32", "mid", 10, 0)
    # ... from 2 KB blocks up (1 KB blocks measured slower with the shorter MID), and not where the outer tile would need address windows
    assert parse(describe(hip_lib, 1 << 19, 2048)[1])[1] == ("T32", "mid", 9, 0)
    assert parse(describe(hip_lib, 1 << 19, 2052)[1])[1] == ("T32", "mid", 9, 0)
    assert parse(describe(hip_lib, 1 << 19, 1024)[1])[1] == ("T32", "mid", 10, 0)
    assert parse(describe(hip_lib, 1 << 19, 8192)[1])[1] == ("T32", "mid", 10, 0)
    rc, text = describe(hip_lib, 1 << 19, 4096, 3100)  # the plan of rounds 1-4, also what the decoder's contexts run
    assert parse(text) == [("S32", "dif", 9, 10), ("T32", "mid", 10, 0), ("S32", "dit", 9, 10)], text
    rc, text = describe(hip_lib, 1 << 19, 4096, 1100)
    assert parse(text) == [("T32", "dif", 9, 10), ("T32", "mid", 10, 0), ("T32", "dit", 9, 10)], text


@pytest.mark.parametrize("log2k", range(1, 20))
@pytest.mark.parametrize("plan", [0, 31, 54, 1060, 1081, 1090, 1100, 2080, 2100, 3090, 3100, 4090, 4100])
def test_every_plan_covers_every_level_once(hip_lib, log2k, plan):
    rc, text = describe(hip_lib, 1 << log2k, 4096, plan)
    assert rc == 0, (log2k, plan)
    passes = parse(text)
    mids = [p for p in passes if p[1] == "mid"]
    assert len(mids) == 1 and mids[0][3] == 0
    down = [p for p in passes if p[1] == "dif"]
    up = [p for p in passes if p[1] == "dit"]
    # DIF passes walk the levels from the top down to the MID block, DIT passes mirror them
    top = log2k
    for kind, _, levels, s in down:
        assert s + levels == top
        top = s
    assert top == mids[0][2]
    assert [(k, l, s) for k, _, l, s in up] == [(k, l, s) for k, _, l, s in reversed(down)]


def test_blocks_too_large_for_tiles_use_register_passes(hip_lib):
    rc, text = describe(hip_lib, 64, (1 << 26) + 16)   # a 64-block tile would span > 2^32 bytes
    assert rc == 0 and "T32" not in text and "T64" not in text and "S32" not in text
    rc, text = describe(hip_lib, 64, (1 << 25) + 16)   # 2^31 < span < 2^32: still addressable with 32-bit offsets
    assert rc == 0 and "T64:mid6@0" in text
    rc, text = describe(hip_lib, 1 << 19, 4100)  # 4096 data bytes + the packing word (GF.md:72-104): still 3 tile trips
    assert rc == 0 and parse(text) == [("T32", "dif", 10, 9), ("T32", "mid", 9, 0), ("T32", "dit", 10, 9)], text
    rc, text = describe(hip_lib, 1 << 19, 65536)  # 64 KB blocks (BASELINE config 5's block size)
    assert rc == 0 and "T32:mid10@0" in text and "T32:dif" not in text, text


def test_plan_argument_validation(hip_lib):
    assert describe(hip_lib, 100, 4096)[0] == -1
    assert describe(hip_lib, 1 << 20, 4096)[0] == -4
    assert describe(hip_lib, 256, 4098)[0] == -1
    assert describe(hip_lib, 256, 4096, 99)[0] == -1
    assert describe(hip_lib, 256, 4096, 1118)[0] == -1


@pytest.mark.parametrize("log2k,plan", [(3, 0), (7, 0), (10, 0), (13, 0), (13, 51), (16, 1090), (18, 0), (19, 0), (19, 1100)])
def test_level_tables_hold_the_reference_roots(hip_lib, oracle, log2k, plan):
    """Entry 2^l + ((i mod 2^sl) << (l-sl)) + (i >> sl) must be (root of order 2^(l+1))^i * 2^32 mod p, with
    the root the reference uses (GF_Root = 19^((p-1)/N), inverse for the interpolation half)."""
    k = 1 << log2k
    rng = np.random.default_rng(log2k)
    for which, inverse in ((0, True), (1, False), (2, False), (3, True)):
        tab, sl = twiddles(hip_lib, k, 4096, plan, which)
        w = oracle.gf_root(k)
        if inverse:
            w = oracle.gf_inv(w)
        for l in range(log2k):
            h = 1 << l
            root = oracle.gf_pow(w, k >> (l + 1))
            idx = set([0, 1, h - 1, h // 2] + rng.integers(0, h, 6).tolist()) if h > 1 else {0}
            for i in idx:
                if i >= h:
                    continue
                t = l - sl[l]
                assert 0 <= sl[l] <= l
                pos = h + (((i & ((1 << sl[l]) - 1)) << t) | (i >> sl[l]))
                want = (oracle.gf_pow(root, i) << 32) % P
                assert int(tab[pos]) == want, (which, l, i)


def test_tile_eligibility_follows_the_block_span(hip_lib):
    """32-bit buffer offsets: an outer tile must span < 2^32 bytes, or up to 2^36 with 2 … 16 address windows
    (W forms); beyond that the top levels fall back to register passes with 64-bit addressing."""
    assert describe(hip_lib, 1 << 19, 4096)[1].startswith("T32:dif10@9,T32:mid9@0,T32:dit10@9")
    assert describe(hip_lib, 1 << 19, 4096, 3100)[1].startswith("S32:dif9@10,T32:mid10@0,S32:dit9@10")
    assert describe(hip_lib, 1 << 19, 8192)[1].startswith("SW32:dif9@10,T32:mid10@0,SW32:dit9@10")
    # narrow blocks in whole 64-word rows at k = 2^19 (the sub-slabs of a stripe over 2 .. 8 GPUs): 64-word-row outer tiles around MID10
    for bb in (256, 512, 1024):
        assert describe(hip_lib, 1 << 19, bb)[1].startswith("T64:dif9@10,T32:mid10@0,T64:dit9@10"), bb
    assert describe(hip_lib, 1 << 19, 128)[1].startswith("S32:dif9@10,T32:mid10@0,S32:dit9@10")     # no 64-word rows
    assert describe(hip_lib, 1 << 19, 384)[1].startswith("S32:dif9@10,T32:mid10@0,S32:dit9@10")     # 96 words: not whole 64-word rows
    assert describe(hip_lib, 1 << 18, 512)[1].startswith("S32:dif8@10,T32:mid10@0,S32:dit8@10")     # the rule is for the 9-level outer chunks
    assert describe(hip_lib, 1 << 18, 16384)[1].startswith("SW32:dif8@10,T32:mid10@0,SW32:dit8@10")
    assert describe(hip_lib, 1 << 19, 16384)[1].startswith("SW4x32:dif9@10,T32:mid10@0,SW4x32:dit9@10")
    assert describe(hip_lib, 1 << 19, 32768)[1].startswith("SW8x32:dif9@10,T32:mid10@0,SW8x32:dit9@10")
    assert describe(hip_lib, 1 << 19, 65536)[1].startswith("SW16x32:dif9@10,T32:mid10@0,SW16x32:dit9@10")
    kinds = [p[0] for p in parse(describe(hip_lib, 1 << 19, 1 << 17)[1])]   # 2^36-byte spans: past sixteen windows of < 2^32
    assert kinds == ["reg", "reg", "T32", "reg", "reg"]
    assert describe(hip_lib, 128, 4096)[1].startswith("T32:mid7@0")
    # MID (contiguous blocks) stays a tile as long as 1024 blocks fit the offsets
    assert parse(describe(hip_lib, 1 << 12, 1 << 21)[1])[1][0] == "T32"
    assert all(p[0] == "reg" for p in parse(describe(hip_lib, 1 << 12, 1 << 23)[1]))
