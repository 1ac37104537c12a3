"""CPU tests of the drop-in boundary: the shared library loads, exports every symbol include/fastecc.h
declares, validates arguments before touching the device, and fails loudly without a GPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "fastecc.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fastecc_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_the_path():
    syms = declared_symbols()
    for must in ("fastecc_create", "fastecc_destroy", "fastecc_encode", "fastecc_encode_blocks", "fastecc_ntt",
                 "fastecc_scale_blocks", "fastecc_gf_binary", "fastecc_strerror"):
        assert must in syms


def test_library_exports_every_declared_symbol(hip_lib):
    for name in declared_symbols():
        assert hasattr(hip_lib, name), "libfastecc_hip.so does not export " + name


def test_host_field_helpers(hip_lib, oracle):
    import fastecc_amd as fe
    P = fe.P
    for x, y in [(0, 5), (1, P - 1), (P - 1, P - 1), (123456789, 987654321)]:
        assert fe.gf_mul(x, y) == oracle.gf_mul(x, y)
    for order in (2, 4, 1 << 10, 1 << 20):
        assert fe.gf_root(order) == oracle.gf_root(order)
    assert fe.gf_root(1 << 21) == 0  # 2^21 does not divide p-1 = 2^20*4095 (GF.md:20)
    assert fe.gf_inv(1 << 19) == oracle.gf_inv(1 << 19)
    assert fe.gf_pow(19, 12345) == oracle.gf_pow(19, 12345)


def test_argument_validation_needs_no_device(hip_lib):
    import fastecc_amd as fe
    h = ctypes.c_void_p()
    create = hip_lib.fastecc_create
    # n <= k, k = 0, block_bytes % 4, zero block, unknown field, k > 2^19, more parity than the transform size (and not 4k / 8k)
    assert create(ctypes.byref(h), 64, 64, 4096, 0, 0) == fe.E_INVAL
    assert create(ctypes.byref(h), 60, 64, 4096, 0, 0) == fe.E_INVAL
    assert create(ctypes.byref(h), 2, 0, 4096, 0, 0) == fe.E_INVAL
    assert create(ctypes.byref(h), 256, 128, 4098, 0, 0) == fe.E_INVAL
    assert create(ctypes.byref(h), 256, 128, 0, 0, 0) == fe.E_INVAL
    assert create(ctypes.byref(h), 256, 128, 4096, 7, 0) == fe.E_UNSUPPORTED
    assert create(ctypes.byref(h), 1 << 21, 1 << 20, 4096, 0, 0) == fe.E_UNSUPPORTED
    assert create(ctypes.byref(h), 100 + 129, 100, 4096, 0, 0) == fe.E_UNSUPPORTED
    assert create(ctypes.byref(h), 3 * 128, 128, 4096, 0, 0) == fe.E_UNSUPPORTED
    assert create(None, 256, 128, 4096, 0, 0) == fe.E_INVAL
    assert not h.value
    assert hip_lib.fastecc_encode(None, None, None, 1, None) == fe.E_INVAL
    assert hip_lib.fastecc_strerror(fe.E_DEVICE).decode().startswith("HIP")
    assert hip_lib.fastecc_version() >= 100


def test_no_cpu_fallback_without_gpu(hip_lib):
    """On a box without a GPU the product must refuse to run rather than compute on the CPU."""
    import torch
    import fastecc_amd as fe
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the -m gpu tests")
    with pytest.raises(fe.FastEccError) as ei:
        fe.Encoder(256, 128, 4096)
    assert ei.value.code == fe.E_DEVICE


def test_product_does_not_reference_the_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline may touch oracle/ (task rule)."""
    pkg = os.path.join(ROOT, "fastecc_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "import oracle" not in text and "from oracle" not in text, f
                assert "fastecc_oracle" not in text and "libfastecc_ref" not in text, f


def test_p61_field_host_helpers_and_validation(hip_lib):
    """FASTECC_FIELD_GF_P61_SQUARED: scalar helpers against the oracle, argument checks before any device use."""
    import fastecc_amd as fe
    import oracle as orc
    o = orc.OracleP61()
    for t in (1, 2, 3, 10, 20, 62):
        assert fe.gf61_root(1 << t) == o.root(1 << t)
    with pytest.raises(fe.FastEccError):
        fe.gf61_root(24)
    x, y = (123456789012345678, 2305843009213693950), (2305843009213693951 + 5, 77)  # y.re is reduced mod p first
    assert fe.gf61_mul(x, y) == o.cmul(x, (5, 77))
    assert fe.gf61_pow(x, (1 << 61) + 12345) == o.cpow(x, (1 << 61) + 12345)
    assert fe.gf61_mul(fe.gf61_inv(x), x) == (1, 0)
    h = ctypes.c_void_p()
    create = hip_lib.fastecc_create
    assert create(ctypes.byref(h), 256, 128, 4100, fe.FIELD_GF_P61_SQUARED, 0) == fe.E_INVAL      # block_bytes % 16
    assert create(ctypes.byref(h), 1 << 26, 1 << 25, 64, fe.FIELD_GF_P61_SQUARED, 0) == fe.E_UNSUPPORTED
    assert not h.value


def test_header_is_plain_c_and_a_c_host_links(tmp_path, hip_lib):
    """include/fastecc.h must be consumable by the bindings INTEGRATION.md shows (cgo, JNI, ctypes are all C): compile a C99
    translation unit that takes the address of every entry point, link it against the library, and run it without a GPU
    (argument validation and the host-side field helpers only)."""
    import subprocess
    from fastecc_amd import _build
    names = declared_symbols()
    src = tmp_path / "host.c"
    src.write_text(
        '#include "fastecc.h"\n#include <stdio.h>\n'
        "int main(void) {\n"
        "    typedef void (*entry_point)(void);\n"
        "    entry_point entry[] = {" + ", ".join("(entry_point)%s" % n for n in names) + "};\n"
        "    fastecc_ctx *ctx = 0;\n"
        "    if (fastecc_create(&ctx, 8, 8, 64, FASTECC_FIELD_GF_FFF00001, 0) != FASTECC_E_INVAL) return 2;  /* n <= k */\n"
        "    if (fastecc_encode(0, 0, 0, FASTECC_MEM_DEVICE, 0) != FASTECC_E_INVAL) return 3;\n"
        "    if (fastecc_gf_mul(2, 3) != 6u || fastecc_version() != FASTECC_VERSION) return 4;\n"
        '    printf("%d entry points\\n", (int)(sizeof entry / sizeof entry[0]));\n'
        "    return 0;\n}\n")
    exe = tmp_path / "host"
    lib_dir = os.path.dirname(_build.LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                    "-L", lib_dir, "-lfastecc_hip", "-Wl,-rpath," + lib_dir, "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"],
                   check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert out.stdout.strip() == "%d entry points" % len(names) and len(names) >= 30
