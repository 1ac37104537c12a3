"""fastecc_amd — MI355X-native NTT Reed-Solomon encode path (FastECC-compatible).

Host-side Python harness over the C ABI of ``fastecc_amd/lib/libfastecc_hip.so`` (include/fastecc.h).
The product is the shared library; this module only binds it with ctypes so that tests and bench.py can
call exactly the entry points a C++ host (fastecc_amd/host/rs_main.cpp, or FastECC's RS.cpp patched as
in INTEGRATION.md) calls.  torch is used by callers for device memory and streams only — no torch type
appears in any signature here, just integer addresses.

There is no CPU compute path in this package: if the HIP library is missing or no GPU is present the
calls raise.
"""
import ctypes
import os

from . import _build

P = 0xFFF00001  # RS.cpp:86
P61 = (1 << 61) - 1
FIELD_GF_FFF00001 = 0
FIELD_GF_P61_SQUARED = 1  # GF((2^61-1)^2), 16-byte elements (re, im): include/fastecc.h
MEM_HOST, MEM_DEVICE, MEM_HOST_PINNED = 0, 1, 2
CODE_MIXED_RADIX = 1  # fastecc_create_ex flag: transform order q * 2^m, q in {1, 3, 5, 7, 9, 13, 15}
CODE_MIXED_RADIX_PFA = 4  # ... and the composite q = 21, 35, 39, 45, 63, 65, 91, 105, 117 (prime-factor map)
CODE_TOP_RADIX2 = 2  # fastecc_create_ex flag (A/B experiment): the top level of a power-of-two transform through the fused odd-radix kernel

OK, E_INVAL, E_NOMEM, E_DEVICE, E_UNSUPPORTED = 0, -1, -2, -3, -4

_LIB = None


class FastEccError(RuntimeError):
    def __init__(self, code, what):
        self.code = code
        detail = lib().fastecc_last_error_detail().decode()
        super().__init__("%s: %s (%d)%s" % (what, lib().fastecc_strerror(code).decode(), code,
                                             " [" + detail + "]" if detail else ""))


def lib_path():
    # FASTECC_HIP_LIB lets experiments A/B another build of the same library (e.g. different hipcc flags)
    return os.environ.get("FASTECC_HIP_LIB") or _build.LIB_PATH


def lib():
    """Load libfastecc_hip.so (never falls back to anything else)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise FileNotFoundError(
            "%s not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(fastecc_amd has no CPU fallback)" % path)
    # One HIP runtime per process: PyTorch wheels bundle their own libamdhip64/libhsa-runtime64, and a
    # process that initialises two HSA runtimes loses the GPU in the second one ("no ROCm-capable
    # device").  Callers of this harness use torch for device memory, so let torch's runtime load first;
    # libfastecc_hip.so's DT_NEEDED libamdhip64.so.* then binds to the copy already in the process.  A
    # C++ host (fastecc_amd/host/rs_main.cpp) links the system runtime directly and has no such issue.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(path)
    vp, u64, u32, i32 = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int
    L.fastecc_strerror.argtypes, L.fastecc_strerror.restype = [i32], ctypes.c_char_p
    L.fastecc_version.argtypes, L.fastecc_version.restype = [], i32
    L.fastecc_last_error_detail.argtypes, L.fastecc_last_error_detail.restype = [], ctypes.c_char_p
    L.fastecc_create.argtypes, L.fastecc_create.restype = [ctypes.POINTER(vp), u64, u64, u64, i32, i32], i32
    L.fastecc_create_ex.argtypes, L.fastecc_create_ex.restype = [ctypes.POINTER(vp), u64, u64, u64, i32, i32, ctypes.c_uint], i32
    L.fastecc_destroy.argtypes, L.fastecc_destroy.restype = [vp], None
    L.fastecc_encode.argtypes, L.fastecc_encode.restype = [vp, vp, vp, i32, vp], i32
    L.fastecc_encode_batch.argtypes, L.fastecc_encode_batch.restype = [vp, vp, vp, u64, vp], i32
    L.fastecc_encode_columns.argtypes, L.fastecc_encode_columns.restype = [vp, vp, vp, u64, u64, vp], i32
    L.fastecc_create_sharded.argtypes = [ctypes.POINTER(vp), u64, u64, u64, i32, ctypes.POINTER(i32), i32]
    L.fastecc_create_sharded.restype = i32
    L.fastecc_encode_sharded.argtypes, L.fastecc_encode_sharded.restype = [vp, ctypes.POINTER(vp), ctypes.POINTER(vp), vp, vp], i32
    L.fastecc_encode_sharded_blocks.argtypes = [vp, ctypes.POINTER(vp), i32, ctypes.POINTER(vp), vp]
    L.fastecc_encode_sharded_blocks.restype = i32
    L.fastecc_shard_info.argtypes = [vp, ctypes.POINTER(i32), ctypes.POINTER(u64), ctypes.POINTER(i32), i32]
    L.fastecc_shard_info.restype = i32
    L.fastecc_plan_describe.argtypes, L.fastecc_plan_describe.restype = [u64, u64, i32, ctypes.c_char_p, ctypes.c_size_t], i32
    L.fastecc_plan_twiddles.argtypes = [u64, u64, i32, i32, ctypes.POINTER(u32), ctypes.POINTER(ctypes.c_int32)]
    L.fastecc_plan_twiddles.restype = i32
    L.fastecc_encode_blocks.argtypes, L.fastecc_encode_blocks.restype = [vp, ctypes.POINTER(vp)], i32
    L.fastecc_ntt.argtypes, L.fastecc_ntt.restype = [vp, vp, i32, i32, vp], i32
    L.fastecc_scale_blocks.argtypes, L.fastecc_scale_blocks.restype = [vp, vp, u32, u32, i32, vp], i32
    L.fastecc_gf_binary.argtypes, L.fastecc_gf_binary.restype = [vp, i32, vp, vp, vp, u64, vp], i32
    L.fastecc_check_range.argtypes, L.fastecc_check_range.restype = [vp, vp, i32, vp, ctypes.POINTER(u64)], i32
    for name in ("mul", "pow"):
        f = getattr(L, "fastecc_gf_" + name)
        f.argtypes, f.restype = [u32, u32], u32
    for name in ("root", "inv"):
        f = getattr(L, "fastecc_gf_" + name)
        f.argtypes, f.restype = [u32], u32
    u8p = ctypes.POINTER(ctypes.c_uint8)
    L.fastecc_decode_prepare.argtypes, L.fastecc_decode_prepare.restype = [vp, u8p, u8p], i32
    L.fastecc_decode.argtypes, L.fastecc_decode.restype = [vp, vp, vp, i32, vp], i32
    L.fastecc_repair.argtypes, L.fastecc_repair.restype = [vp, vp, vp, i32, vp], i32
    L.fastecc_pack_blocks.argtypes, L.fastecc_pack_blocks.restype = [vp, vp, vp, i32, vp], i32
    L.fastecc_unpack_blocks.argtypes, L.fastecc_unpack_blocks.restype = [vp, vp, vp, i32, vp, ctypes.POINTER(u64)], i32
    pair = ctypes.POINTER(u64)
    L.fastecc_gf61_mul.argtypes, L.fastecc_gf61_mul.restype = [pair, pair, pair], i32
    L.fastecc_gf61_pow.argtypes, L.fastecc_gf61_pow.restype = [pair, u64, pair], i32
    L.fastecc_gf61_inv.argtypes, L.fastecc_gf61_inv.restype = [pair, pair], i32
    L.fastecc_gf61_root.argtypes, L.fastecc_gf61_root.restype = [u64, pair], i32
    L.fastecc_profile_enable.argtypes, L.fastecc_profile_enable.restype = [vp, i32], i32
    L.fastecc_profile_reset.argtypes, L.fastecc_profile_reset.restype = [vp], i32
    L.fastecc_profile_read.argtypes = [vp, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_double),
                                       ctypes.POINTER(u64), i32]
    L.fastecc_profile_read.restype = i32
    L.fastecc_profile_read_bytes.argtypes = [vp, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_double),
                                             ctypes.POINTER(u64), ctypes.POINTER(u64), i32]
    L.fastecc_profile_read_bytes.restype = i32
    L.fastecc_set_option.argtypes, L.fastecc_set_option.restype = [vp, ctypes.c_char_p, i32], i32
    L.fastecc_plan_string.argtypes, L.fastecc_plan_string.restype = [vp], ctypes.c_char_p
    L.fastecc_set_plan.argtypes, L.fastecc_set_plan.restype = [vp, i32], i32
    _LIB = L
    return L


def _check(code, what):
    if code < 0:
        raise FastEccError(code, what)
    return code


def _addr(x):
    """Integer address of a torch tensor / numpy array / int."""
    if x is None:
        return None
    if isinstance(x, int):
        return x
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    if hasattr(x, "ctypes"):
        return x.ctypes.data
    raise TypeError("need an address, torch tensor or numpy array")


class Encoder:
    """(n,k) Reed-Solomon encoder: the RS.cpp:22-68 operation behind the C ABI (include/fastecc.h).

    ``data`` is a device tensor (or raw address) of k*block_bytes bytes laid out block-major, exactly the
    ``T** data`` stripe of RS.cpp:28-33 stored back to back; ``parity`` holds n-k blocks the same way.
    (n,k) = (2N,N), N = 2^m, is the reference's configuration (parity block j = f(w_2N^(2j+1))); fastecc_create
    also accepts n-k = k/2 .. k/16 (a sub-coset of that parity), n = 4k / 8k (further cosets) and any other
    (n,k) with n-k <= 2^ceil(log2 k) by zero extension, all over GF(0xFFF00001) with 4-byte words; and
    (2N,N) over GF((2^61-1)^2) with 16-byte elements (``field=FIELD_GF_P61_SQUARED``).
    """

    def __init__(self, n, k, block_bytes, device=0, field=FIELD_GF_FFF00001, flags=0):
        self._h = ctypes.c_void_p()
        self.n, self.k, self.block_bytes, self.device, self.field = n, k, block_bytes, device, field
        if flags:
            _check(lib().fastecc_create_ex(ctypes.byref(self._h), n, k, block_bytes, field, device, flags), "fastecc_create_ex")
        else:
            _check(lib().fastecc_create(ctypes.byref(self._h), n, k, block_bytes, field, device), "fastecc_create")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            lib().fastecc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def parity_blocks(self):
        return self.n - self.k

    @property
    def words_per_block(self):
        return self.block_bytes // 4

    def encode(self, data, parity=None, stream=0, mem=MEM_DEVICE):
        """parity <- encode(data); parity=None encodes in place (the reference's behaviour)."""
        if parity is None:
            parity = data
        _check(lib().fastecc_encode(self._h, _addr(data), _addr(parity), mem, stream or None), "fastecc_encode")
        return parity

    def encode_columns(self, data, parity, col0_words, width_words, stream=0):
        """Encode only words [col0, col0+width) of every block (columns are independent transforms)."""
        _check(lib().fastecc_encode_columns(self._h, _addr(data), _addr(parity), col0_words, width_words, stream or None),
               "fastecc_encode_columns")
        return parity

    def encode_batch(self, data, parity, count, stream=0):
        """`count` stripes stored back to back in device memory, one launch per pass."""
        if parity is None:
            parity = data
        _check(lib().fastecc_encode_batch(self._h, _addr(data), _addr(parity), count, stream or None), "fastecc_encode_batch")
        return parity

    def encode_host(self, data_np, parity_np=None):
        return self.encode(data_np, parity_np, mem=MEM_HOST)

    def encode_blocks(self, block_addresses):
        arr = (ctypes.c_void_p * len(block_addresses))(*block_addresses)
        _check(lib().fastecc_encode_blocks(self._h, arr), "fastecc_encode_blocks")

    def ntt(self, data, inverse=False, stream=0, mem=MEM_DEVICE):
        _check(lib().fastecc_ntt(self._h, _addr(data), int(inverse), mem, stream or None), "fastecc_ntt")
        return data

    def scale_blocks(self, data, scale, base, stream=0, mem=MEM_DEVICE):
        _check(lib().fastecc_scale_blocks(self._h, _addr(data), scale, base, mem, stream or None),
               "fastecc_scale_blocks")
        return data

    def gf_binary(self, op, x, y, out, count, stream=0):
        code = {"add": 0, "sub": 1, "mul": 2, "mul_mont": 3}[op] if isinstance(op, str) else op
        _check(lib().fastecc_gf_binary(self._h, code, _addr(x), _addr(y), _addr(out), count, stream or None),
               "fastecc_gf_binary")
        return out

    def check_range(self, data, stream=0, mem=MEM_DEVICE):
        """Number of words >= p in the stripe (0 = encodable); README.md:160-162 of the reference."""
        bad = ctypes.c_uint64()
        _check(lib().fastecc_check_range(self._h, _addr(data), mem, stream or None, ctypes.byref(bad)), "fastecc_check_range")
        return int(bad.value)

    def decode_prepare(self, data_present, parity_present):
        """Erasure pattern: k data flags and n - k parity flags (truthy = the block survives)."""
        if len(data_present) != self.k or len(parity_present) != self.n - self.k:
            raise ValueError("need k data flags and n - k parity flags")
        def flags(v, count):
            # a contiguous uint8 numpy array goes through as it is (the C ABI takes plain byte arrays); anything else is converted
            if hasattr(v, "ctypes") and getattr(v, "dtype", None) is not None and v.dtype.itemsize == 1 and v.flags["C_CONTIGUOUS"]:
                return v, ctypes.cast(v.ctypes.data, ctypes.POINTER(ctypes.c_uint8))
            arr = (ctypes.c_uint8 * count)(*[1 if x else 0 for x in v])
            return arr, arr
        keep_d, dp = flags(data_present, self.k)
        keep_p, pp = flags(parity_present, self.n - self.k)
        _check(lib().fastecc_decode_prepare(self._h, dp, pp), "fastecc_decode_prepare")
        del keep_d, keep_p

    def decode(self, data, parity, stream=0, mem=MEM_DEVICE):
        """Recover the erased data blocks in place (README.md:102-119); parity is read only."""
        _check(lib().fastecc_decode(self._h, _addr(data), _addr(parity), mem, stream or None), "fastecc_decode")
        return data

    def repair(self, data, parity, stream=0, mem=MEM_DEVICE):
        """decode, then rebuild the erased parity blocks as well (both buffers are written where blocks were lost)."""
        _check(lib().fastecc_repair(self._h, _addr(data), _addr(parity), mem, stream or None), "fastecc_repair")
        return data, parity

    def pack_blocks(self, raw, packed, stream=0, mem=MEM_DEVICE):
        """GF.md:72-104: k blocks of block_bytes - 4 arbitrary bytes -> k encodable blocks of block_bytes."""
        _check(lib().fastecc_pack_blocks(self._h, _addr(raw), _addr(packed), mem, stream or None), "fastecc_pack_blocks")
        return packed

    def unpack_blocks(self, packed, raw, stream=0, mem=MEM_DEVICE, count_bad=True):
        """Inverse of pack_blocks; returns the number of blocks that are not packer output (None if not counted)."""
        bad = ctypes.c_uint64()
        _check(lib().fastecc_unpack_blocks(self._h, _addr(packed), _addr(raw), mem, stream or None,
                                           ctypes.byref(bad) if count_bad else None), "fastecc_unpack_blocks")
        return int(bad.value) if count_bad else None

    # ---- introspection used by bench.py ----
    def set_plan(self, plan):
        _check(lib().fastecc_set_plan(self._h, plan), "fastecc_set_plan")

    def plan(self):
        return lib().fastecc_plan_string(self._h).decode()

    def profile(self, on=True):
        _check(lib().fastecc_profile_enable(self._h, int(on)), "fastecc_profile_enable")

    def profile_reset(self):
        _check(lib().fastecc_profile_reset(self._h), "fastecc_profile_reset")

    def set_option(self, name, value):
        _check(lib().fastecc_set_option(self._h, name.encode(), int(value)), "fastecc_set_option")

    def profile_read(self, cap=64):
        """{kernel: (total ms, launches, total algorithmic bytes)} since the last reset."""
        names = (ctypes.c_char_p * cap)()
        ms = (ctypes.c_double * cap)()
        cnt = (ctypes.c_uint64 * cap)()
        nbytes = (ctypes.c_uint64 * cap)()
        n = _check(lib().fastecc_profile_read_bytes(self._h, names, ms, cnt, nbytes, cap), "fastecc_profile_read_bytes")
        return {names[i].decode(): (ms[i], int(cnt[i]), int(nbytes[i])) for i in range(n)}


class ShardedEncoder(Encoder):
    """One stripe in column slabs on several GPUs driven by ONE process (fastecc_create_sharded): slab g =
    words [g*S/G, (g+1)*S/G) of every block lives on ``gpu_ids[g]``; ``gpu_ids[0]`` is the root that holds full
    stripes.  ``encode(data, parity)`` takes full stripes (root device or host memory);
    ``encode_sharded(data_slabs, parity_slabs=None, parity=None)`` takes data that is already sharded."""

    def __init__(self, n, k, block_bytes, gpu_ids, field=FIELD_GF_FFF00001):
        self._h = ctypes.c_void_p()
        self.n, self.k, self.block_bytes, self.field = n, k, block_bytes, field
        self.gpu_ids = list(gpu_ids)
        self.device = self.gpu_ids[0] if self.gpu_ids else 0
        ids = (ctypes.c_int * len(self.gpu_ids))(*self.gpu_ids)
        _check(lib().fastecc_create_sharded(ctypes.byref(self._h), n, k, block_bytes, field, ids, len(self.gpu_ids)),
               "fastecc_create_sharded")

    @property
    def slab_block_bytes(self):
        return self.block_bytes // len(self.gpu_ids)

    def encode_sharded(self, data_slabs, parity_slabs=None, parity=None, stream=0):
        g = len(self.gpu_ids)
        d = (ctypes.c_void_p * g)(*[_addr(x) for x in data_slabs])
        p = (ctypes.c_void_p * g)(*[_addr(x) for x in parity_slabs]) if parity_slabs is not None else None
        _check(lib().fastecc_encode_sharded(self._h, d, p, _addr(parity), stream or None), "fastecc_encode_sharded")
        return parity if parity is not None else parity_slabs

    def encode_sharded_blocks(self, data, parity_blocks, data_is_blocks=False, stream=0):
        """Block-distributed result: parity_blocks[g] receives parity blocks [g*M/G, (g+1)*M/G) whole on GPU g (all-to-all over the peers).
        data[g]: column slab g, or with data_is_blocks the data blocks [g*k/G, (g+1)*k/G) whole."""
        g = len(self.gpu_ids)
        d = (ctypes.c_void_p * g)(*[_addr(x) for x in data])
        p = (ctypes.c_void_p * g)(*[_addr(x) for x in parity_blocks])
        _check(lib().fastecc_encode_sharded_blocks(self._h, d, 1 if data_is_blocks else 0, p, stream or None), "fastecc_encode_sharded_blocks")
        return parity_blocks


MIXED_RADIX_Q = (1, 3, 5, 7, 9, 13, 15)
MIXED_RADIX_PFA_Q = MIXED_RADIX_Q + (21, 35, 39, 45, 63, 65, 91, 105, 117)


def mixed_radix_order(k, pfa=False):
    """Transform order fastecc_create_ex(..., CODE_MIXED_RADIX) picks for k data blocks: the smallest q * 2^m >= k,
    q in {1, 3, 5, 7, 9, 13, 15} (pfa: CODE_MIXED_RADIX_PFA, also 21 ... 117), 1 <= m <= 19 (None if there is none)."""
    best = None
    for q in (MIXED_RADIX_PFA_Q if pfa else MIXED_RADIX_Q):
        for m in range(1, 20):
            if (q << m) >= k and (best is None or (q << m) < best):
                best = q << m
    return best


def plan_describe(k, block_bytes, plan=0):
    """Host-only: the pass plan fastecc_create would pick (no device is touched)."""
    buf = ctypes.create_string_buffer(512)
    _check(lib().fastecc_plan_describe(k, block_bytes, plan, buf, 512), "fastecc_plan_describe")
    return buf.value.decode()


def plan_twiddles(k, block_bytes, plan=0, which=0):
    """Host-only: (level-packed twiddle table as a list of k ints, per-level log2 strides)."""
    n = max(k.bit_length() - 1, 0)
    out = (ctypes.c_uint32 * k)()
    sl = (ctypes.c_int32 * max(n, 1))()
    _check(lib().fastecc_plan_twiddles(k, block_bytes, plan, which, out, sl), "fastecc_plan_twiddles")
    return list(out), list(sl)[:n]


def gf_mul(x, y): return lib().fastecc_gf_mul(x, y)
def gf_pow(x, e): return lib().fastecc_gf_pow(x, e)
def gf_root(order): return lib().fastecc_gf_root(order)
def gf_inv(x): return lib().fastecc_gf_inv(x)


# GF((2^61-1)^2) scalars, (re, im) tuples
def _pair(z):
    return (ctypes.c_uint64 * 2)(int(z[0]), int(z[1]))


def gf61_mul(x, y):
    out = _pair((0, 0))
    _check(lib().fastecc_gf61_mul(_pair(x), _pair(y), out), "fastecc_gf61_mul")
    return (int(out[0]), int(out[1]))


def gf61_pow(x, e):
    out = _pair((0, 0))
    _check(lib().fastecc_gf61_pow(_pair(x), e, out), "fastecc_gf61_pow")
    return (int(out[0]), int(out[1]))


def gf61_inv(x):
    out = _pair((0, 0))
    _check(lib().fastecc_gf61_inv(_pair(x), out), "fastecc_gf61_inv")
    return (int(out[0]), int(out[1]))


def gf61_root(order):
    out = _pair((0, 0))
    _check(lib().fastecc_gf61_root(order, out), "fastecc_gf61_root")
    return (int(out[0]), int(out[1]))
