"""CPU oracle bindings — TEST INFRASTRUCTURE ONLY.

ctypes views of oracle/libfastecc_oracle.so (our plain-C restatement, fastecc_oracle.c) and, when it
has been built, oracle/_ref/libfastecc_ref*.so (the unmodified reference behind ref_shim.cpp).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the
product package (fastecc_amd) never does.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
P = 0xFFF00001

_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_sz = ctypes.c_size_t
_u32 = ctypes.c_uint32


def build(force=False):
    """Compile the restatement (and the reference shim when /root/reference is present)."""
    so = os.path.join(HERE, "libfastecc_oracle.so")
    srcs = [os.path.join(HERE, f) for f in ("fastecc_oracle.c", "fastecc_oracle_p61.c")]
    have_ref = os.path.exists("/root/reference/ntt.cpp")
    stale = (not os.path.exists(so)) or os.path.getmtime(so) < max(os.path.getmtime(s) for s in srcs)
    ref_so, shim = os.path.join(HERE, "_ref", "libfastecc_ref.so"), os.path.join(HERE, "ref_shim.cpp")
    ref_missing = have_ref and (not os.path.exists(ref_so) or os.path.getmtime(ref_so) < os.path.getmtime(shim))  # absent, or older than the shim
    if force or stale or ref_missing:
        subprocess.run(["make", "-C", HERE] + (["-B"] if force else []), check=True,
                       stdout=subprocess.DEVNULL)
    return so


class Oracle:
    """Plain-C restatement (fastecc_oracle.c)."""

    def __init__(self):
        self.lib = lib = ctypes.CDLL(build())
        for name in ("add", "sub", "mul", "mul_wide", "pow"):
            f = getattr(lib, "orc_gf_" + name)
            f.argtypes, f.restype = [_u32, _u32], _u32
        for name in ("root", "inv"):
            f = getattr(lib, "orc_gf_" + name)
            f.argtypes, f.restype = [_u32], _u32
        for name in ("slow_ntt", "ntt", "ntt_fast", "ntt_mixed"):
            f = getattr(lib, "orc_" + name)
            f.argtypes, f.restype = [_u32p, _sz, _sz, ctypes.c_int], None
        lib.orc_scale_blocks.argtypes, lib.orc_scale_blocks.restype = [_u32p, _sz, _sz, _u32, _u32], None
        for name in ("encode", "encode_fast", "encode_slow", "encode_mixed"):
            f = getattr(lib, "orc_" + name)
            f.argtypes, f.restype = [_u32p, _sz, _sz], None
        lib.orc_encode_by_definition.argtypes = [_u32p, _u32p, _sz, _sz]
        lib.orc_encode_by_definition.restype = None
        lib.orc_hash.argtypes, lib.orc_hash.restype = [_u32p, _sz], _u32
        lib.orc_fill_linear.argtypes, lib.orc_fill_linear.restype = [_u32p, _sz], None
        lib.orc_fill_splitmix.argtypes, lib.orc_fill_splitmix.restype = [_u32p, _sz, ctypes.c_uint64], None
        lib.orc_num_threads.restype = ctypes.c_int
        _u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
        lib.orc_decode.argtypes, lib.orc_decode.restype = [_u32p, _u32p, _u8p, _u8p, _sz, _sz], ctypes.c_int
        lib.orc_pack_blocks.argtypes, lib.orc_pack_blocks.restype = [_u32p, _sz, _sz, _u32p], None
        lib.orc_unpack_blocks.argtypes, lib.orc_unpack_blocks.restype = [_u32p, _sz, _sz, _u32p], _sz

    # field
    def gf_add(self, x, y): return self.lib.orc_gf_add(x, y)
    def gf_sub(self, x, y): return self.lib.orc_gf_sub(x, y)
    def gf_mul(self, x, y): return self.lib.orc_gf_mul(x, y)
    def gf_mul_wide(self, x, y): return self.lib.orc_gf_mul_wide(x, y)
    def gf_pow(self, x, n): return self.lib.orc_gf_pow(x, n)
    def gf_root(self, n): return self.lib.orc_gf_root(n)
    def gf_inv(self, x): return self.lib.orc_gf_inv(x)

    # transforms on a [N, size] uint32 array; all return a new array
    def _run(self, fn, data, *extra):
        a = np.ascontiguousarray(data, dtype=np.uint32).copy()
        N, size = a.shape
        fn(a, N, size, *extra)
        return a

    def slow_ntt(self, data, inverse=False): return self._run(self.lib.orc_slow_ntt, data, int(inverse))
    def ntt(self, data, inverse=False): return self._run(self.lib.orc_ntt, data, int(inverse))
    def ntt_fast(self, data, inverse=False): return self._run(self.lib.orc_ntt_fast, data, int(inverse))
    def scale_blocks(self, data, scale, base): return self._run(self.lib.orc_scale_blocks, data, scale, base)
    def encode(self, data): return self._run(self.lib.orc_encode, data)
    def encode_fast(self, data): return self._run(self.lib.orc_encode_fast, data)
    # any transform order N | p-1 (mixed radix, N = q 2^m): by definition, and with the odd factor outermost
    def encode_slow(self, data): return self._run(self.lib.orc_encode_slow, data)
    def encode_mixed(self, data): return self._run(self.lib.orc_encode_mixed, data)
    def ntt_mixed(self, data, inverse=False): return self._run(self.lib.orc_ntt_mixed, data, int(inverse))

    def encode_mixed_code(self, data, n, order):
        """The (n,k) code FASTECC_CODE_MIXED_RADIX builds on transform order `order` >= k: zero-extend, encode, truncate."""
        k, size = data.shape
        x = np.zeros((order, size), dtype=np.uint32)
        x[:k] = data
        return self.encode_mixed(x)[: n - k]

    def encode_fast_inplace(self, a):
        """In-place variant for large buffers (no copy)."""
        N, size = a.shape
        self.lib.orc_encode_fast(a, N, size)
        return a

    def encode_by_definition(self, data):
        a = np.ascontiguousarray(data, dtype=np.uint32)
        out = np.empty_like(a)
        self.lib.orc_encode_by_definition(a, out, a.shape[0], a.shape[1])
        return out

    def hash(self, data):
        a = np.ascontiguousarray(data, dtype=np.uint32).reshape(-1)
        return int(self.lib.orc_hash(a, a.size))

    def fill_linear(self, N, size):
        a = np.empty((N, size), dtype=np.uint32)
        self.lib.orc_fill_linear(a.reshape(-1), a.size)
        return a

    def fill_splitmix(self, N, size, seed=0x1234):
        a = np.empty((N, size), dtype=np.uint32)
        self.lib.orc_fill_splitmix(a.reshape(-1), a.size, seed)
        return a

    def num_threads(self): return int(self.lib.orc_num_threads())

    def decode(self, data, parity, data_present, parity_present):
        """Erased data blocks recomputed by Lagrange interpolation (O(N^2)); returns the repaired data or None."""
        d = np.ascontiguousarray(data, dtype=np.uint32).copy()
        p = np.ascontiguousarray(parity, dtype=np.uint32)
        rc = self.lib.orc_decode(d, p, np.ascontiguousarray(data_present, dtype=np.uint8),
                                 np.ascontiguousarray(parity_present, dtype=np.uint8), d.shape[0], d.shape[1])
        return d if rc == 0 else None

    # data packing (GF.md:72-104): [N, words] arbitrary uint32 <-> [N, words + 1] uint32 < P
    def pack_blocks(self, raw):
        a = np.ascontiguousarray(raw, dtype=np.uint32)
        out = np.empty((a.shape[0], a.shape[1] + 1), dtype=np.uint32)
        self.lib.orc_pack_blocks(a, a.shape[0], a.shape[1], out)
        return out

    def unpack_blocks(self, packed):
        """-> (raw, number of blocks that no packer produces)"""
        a = np.ascontiguousarray(packed, dtype=np.uint32)
        out = np.empty((a.shape[0], a.shape[1] - 1), dtype=np.uint32)
        bad = self.lib.orc_unpack_blocks(a, a.shape[0], a.shape[1] - 1, out)
        return out, int(bad)


P61 = (1 << 61) - 1
_u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
_u64 = ctypes.c_uint64
_pair = ctypes.c_uint64 * 2


class OracleP61:
    """Plain-C restatement of the encode composition over GF((2^61-1)^2) (fastecc_oracle_p61.c).

    PARITY UNPINNED: the reference has no implementation of this field (see fastecc_oracle_p61.h).
    Stripes are [N, 2*elems] uint64 arrays: element c of a block is (re, im) = words (2c, 2c+1).
    """

    def __init__(self):
        self.lib = lib = ctypes.CDLL(build())
        for name in ("add", "sub", "mul"):
            f = getattr(lib, "orc61_" + name)
            f.argtypes, f.restype = [_u64, _u64], _u64
        lib.orc61c_mul.argtypes, lib.orc61c_mul.restype = [_pair, _pair, _pair], None
        lib.orc61c_pow.argtypes, lib.orc61c_pow.restype = [_pair, _u64, _pair], None
        lib.orc61c_inv.argtypes, lib.orc61c_inv.restype = [_pair, _pair], None
        lib.orc61c_root.argtypes, lib.orc61c_root.restype = [_u64, _pair], None
        for name in ("slow_ntt", "ntt"):
            f = getattr(lib, "orc61_" + name)
            f.argtypes, f.restype = [_u64p, _sz, _sz, ctypes.c_int], None
        lib.orc61_scale_blocks.argtypes, lib.orc61_scale_blocks.restype = [_u64p, _sz, _sz, _pair, _pair], None
        lib.orc61_encode.argtypes, lib.orc61_encode.restype = [_u64p, _sz, _sz], None
        lib.orc61_encode_by_definition.argtypes, lib.orc61_encode_by_definition.restype = [_u64p, _u64p, _sz, _sz], None
        lib.orc61_fill_splitmix.argtypes, lib.orc61_fill_splitmix.restype = [_u64p, _sz, _u64], None
        _u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
        lib.orc61_decode.argtypes, lib.orc61_decode.restype = [_u64p, _u64p, _u8p, _u8p, _sz, _sz], ctypes.c_int

    def cmul(self, x, y):
        out = _pair()
        self.lib.orc61c_mul(_pair(*x), _pair(*y), out)
        return (int(out[0]), int(out[1]))

    def cpow(self, x, e):
        out = _pair()
        self.lib.orc61c_pow(_pair(*x), e, out)
        return (int(out[0]), int(out[1]))

    def cinv(self, x):
        out = _pair()
        self.lib.orc61c_inv(_pair(*x), out)
        return (int(out[0]), int(out[1]))

    def root(self, order):
        out = _pair()
        self.lib.orc61c_root(order, out)
        return (int(out[0]), int(out[1]))

    def _run(self, fn, data, *extra):
        a = np.ascontiguousarray(data, dtype=np.uint64).copy()
        fn(a, a.shape[0], a.shape[1] // 2, *extra)
        return a

    def slow_ntt(self, data, inverse=False): return self._run(self.lib.orc61_slow_ntt, data, int(inverse))
    def ntt(self, data, inverse=False): return self._run(self.lib.orc61_ntt, data, int(inverse))
    def encode(self, data): return self._run(self.lib.orc61_encode, data)

    def scale_blocks(self, data, scale, base):
        return self._run(self.lib.orc61_scale_blocks, data, _pair(*scale), _pair(*base))

    def decode(self, data, parity, data_present, parity_present):
        """O(N^2) Lagrange erasure decoding; returns the repaired data stripe."""
        a = np.ascontiguousarray(data, dtype=np.uint64).copy()
        par = np.ascontiguousarray(parity, dtype=np.uint64)
        dp = np.ascontiguousarray(data_present, dtype=np.uint8)
        pp = np.ascontiguousarray(parity_present, dtype=np.uint8)
        if self.lib.orc61_decode(a, par, dp, pp, a.shape[0], a.shape[1] // 2) != 0:
            raise ValueError("fewer than N blocks survive")
        return a

    def encode_by_definition(self, data):
        a = np.ascontiguousarray(data, dtype=np.uint64)
        out = np.empty_like(a)
        self.lib.orc61_encode_by_definition(a, out, a.shape[0], a.shape[1] // 2)
        return out

    def fill_splitmix(self, N, elems, seed=0x1234):
        a = np.empty((N, 2 * elems), dtype=np.uint64)
        self.lib.orc61_fill_splitmix(a.reshape(-1), a.size, seed)
        return a


class Reference:
    """The unmodified reference behind oracle/ref_shim.cpp (only if oracle/_ref/ was built)."""

    @staticmethod
    def path(avx2=False):
        return os.path.join(HERE, "_ref", "libfastecc_ref_avx2.so" if avx2 else "libfastecc_ref.so")

    @classmethod
    def available(cls, avx2=False):
        return os.path.exists(cls.path(avx2))

    def __init__(self, avx2=False):
        self.lib = lib = ctypes.CDLL(self.path(avx2))
        for name in ("add", "sub", "mul", "pow"):
            f = getattr(lib, "ref_gf_" + name)
            f.argtypes, f.restype = [_u32, _u32], _u32
        for name in ("root", "inv"):
            f = getattr(lib, "ref_gf_" + name)
            f.argtypes, f.restype = [_u32], _u32
        lib.ref_ntt.argtypes, lib.ref_ntt.restype = [_u32p, _sz, _sz, ctypes.c_int, ctypes.c_int], None
        lib.ref_encode.argtypes, lib.ref_encode.restype = [_u32p, _sz, _sz], None
        lib.ref_hash.argtypes, lib.ref_hash.restype = [_u32p, _sz, _sz], _u32
        if hasattr(lib, "ref_small_ntt"):
            lib.ref_small_ntt.argtypes, lib.ref_small_ntt.restype = [_u32p, ctypes.c_int, ctypes.c_int], ctypes.c_int
        lib.ref_simd_level.restype = ctypes.c_int

    def gf_add(self, x, y): return self.lib.ref_gf_add(x, y)
    def gf_sub(self, x, y): return self.lib.ref_gf_sub(x, y)
    def gf_mul(self, x, y): return self.lib.ref_gf_mul(x, y)
    def gf_pow(self, x, n): return self.lib.ref_gf_pow(x, n)
    def gf_root(self, n): return self.lib.ref_gf_root(n)
    def gf_inv(self, x): return self.lib.ref_gf_inv(x)

    MFA, REC, SLOW = 0, 1, 2

    def ntt(self, data, inverse=False, which=0):
        a = np.ascontiguousarray(data, dtype=np.uint32).copy()
        self.lib.ref_ntt(a, a.shape[0], a.shape[1], int(inverse), which)
        return a

    def encode(self, data):
        a = np.ascontiguousarray(data, dtype=np.uint32).copy()
        self.lib.ref_encode(a, a.shape[0], a.shape[1])
        return a

    def small_ntt(self, f, inverse=False):
        """The reference's NTT2 / NTT4 (ntt.cpp:16-22, 50-62) or NTT3 / NTT9 codelet (ntt.cpp:25-44, 113-146) on a vector of 2, 4, 3 or 9 words."""
        a = np.ascontiguousarray(f, dtype=np.uint32).copy()
        if self.lib.ref_small_ntt(a, a.size, int(inverse)) != 0:
            raise ValueError("the reference has codelets of order 2, 3, 4 and 9 only (or oracle/_ref predates the order-2 / order-4 entries)")
        return a

    def encode_inplace(self, a):
        self.lib.ref_encode(a, a.shape[0], a.shape[1])
        return a

    def hash(self, data):
        a = np.ascontiguousarray(data, dtype=np.uint32)
        if a.ndim == 1:
            a = a.reshape(1, -1)
        return int(self.lib.ref_hash(a, a.shape[0], a.shape[1]))

    def simd_level(self): return int(self.lib.ref_simd_level())
