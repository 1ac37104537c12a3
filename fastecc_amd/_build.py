"""Build recipe for the gfx950 shared library and the C++ host driver (explicit hipcc, in-tree outputs)."""
import hashlib
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libfastecc_hip.so")
RS_PATH = os.path.join(LIB_DIR, "rs_hip")
MICROBENCH_PATH = os.path.join(LIB_DIR, "microbench")

HIP_SOURCES = ["plan.hip", "options.hip", "kernels.hip", "tile_kernels.hip", "mixed_kernels.hip", "mixed_kernels_pfa.hip", "mixed_kernels_pfa2.hip", "mixed_kernels_pfa3.hip", "gf61_kernels.hip", "gf61_decode.hip", "pack_kernels.hip", "direct.hip", "decode.hip", "sharded.hip", "host_copy.hip", "encode.hip", "host_stage.hip", "create.hip", "api.hip"]
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# experiments only (e.g. -DFASTECC_DIRECT_ABLATION: timing ablations of direct.hip, wrong results on purpose): extra flags from the environment
HIP_FLAGS += os.environ.get("FASTECC_EXTRA_HIPFLAGS", "").split()


FLAGS_STAMP = os.path.join(LIB_DIR, ".hip_flags")
# the sources the committed counter files under profiles/ describe (bench.py labels them STALE when the tree has moved on)
COUNTED_SOURCES = ["tile_kernels.hip", "gf.hpp", "ntt_device.hpp", "kernels.hpp"]


def flags_key(flags=None):
    """What the objects in fastecc_amd/lib were compiled with: objects are reused only under the same flag set, so an experiment's build
    (FASTECC_EXTRA_HIPFLAGS, e.g. the direct path's timing ablations with wrong results on purpose) can never be linked into a later plain one."""
    return hashlib.sha256(" ".join(HIP_FLAGS if flags is None else flags).encode()).hexdigest()


def flags_changed(stamp_path=None, flags=None):
    """True when the objects on disk were built with other flags than the current ones (or nobody recorded theirs)."""
    try:
        with open(stamp_path or FLAGS_STAMP) as f:
            return f.read().strip() != flags_key(flags)
    except OSError:
        return True


def kernel_sources_sha256(csrc=None, names=None):
    """sha256 over the headline kernels' sources, file by file and combined (tools/stamp_counters.py writes it next to the counter files)."""
    per_file, h = {}, hashlib.sha256()
    for name in names or COUNTED_SOURCES:
        with open(os.path.join(csrc or CSRC, name), "rb") as f:
            data = f.read()
        per_file[name] = hashlib.sha256(data).hexdigest()
        h.update(name.encode() + b"\0" + data)
    return {"sha256": h.hexdigest(), "files": per_file}


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found")
    return exe


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _deps():
    out = [os.path.join(ROOT, "include", "fastecc.h")]
    for d in (CSRC, os.path.join(PKG, "host")):
        for f in os.listdir(d):
            out.append(os.path.join(d, f))
    return out


def build_library(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> fastecc_amd/lib/libfastecc_hip.so"""
    os.makedirs(LIB_DIR, exist_ok=True)
    if any(os.path.exists(os.path.join(LIB_DIR, src.replace(".hip", ".o"))) for src in HIP_SOURCES) and flags_changed():
        force = True  # objects of another flag set (or of unknown origin) are never reused
    if not (force or _newer(LIB_PATH, _deps())):
        return LIB_PATH
    objs = []
    headers = [os.path.join(ROOT, "include", "fastecc.h")] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    jobs = []
    for src in HIP_SOURCES:
        obj = os.path.join(LIB_DIR, src.replace(".hip", ".o"))
        objs.append(obj)
        if not (force or _newer(obj, headers + [os.path.join(CSRC, src)])):
            continue  # this translation unit is up to date
        jobs.append([hipcc()] + HIP_FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj])
    if jobs:  # the translation units are independent: compile them side by side
        from concurrent.futures import ThreadPoolExecutor

        def compile_one(cmd):
            if verbose:
                print(" ".join(cmd))
            subprocess.run(cmd, check=True)

        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as pool:
            list(pool.map(compile_one, jobs))
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    with open(FLAGS_STAMP, "w") as f:
        f.write(flags_key() + "\n")
    return LIB_PATH


def build_host(force=False, verbose=False):
    """The rs-compatible C++ host driver (fastecc_amd/host/rs_main.cpp) linked against the C ABI."""
    src = os.path.join(PKG, "host", "rs_main.cpp")
    if not os.path.exists(src):
        return None
    if not (force or _newer(RS_PATH, [src, LIB_PATH])):
        return RS_PATH
    cmd = [hipcc(), "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), src, "-o", RS_PATH,
           "-L", LIB_DIR, "-lfastecc_hip", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return RS_PATH


# the matrix-core prototypes keep their accumulators in VGPRs (the VALU reads them; the AGPR form costs a v_accvgpr_read per word)
MICROBENCH_FLAGS = {"microbench_mfma_dft": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "proto_mid_mfma": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def build_microbench(force=False, verbose=False):
    """tools/microbench*.hip -> fastecc_amd/lib/microbench* (design probes, not part of the library)."""
    for name in ("microbench", "microbench_valu2", "microbench_p61", "microbench_mfma", "microbench_f64", "proto_mid_f64", "microbench_mfma_dft", "proto_mid_mfma", "microbench_mfma_valu_overlap"):
        src = os.path.join(ROOT, "tools", name + ".hip")
        out = os.path.join(LIB_DIR, name)
        if not os.path.exists(src):
            continue
        if not (force or _newer(out, [src, os.path.join(CSRC, "gf.hpp"), os.path.join(CSRC, "gf61.hpp")])):
            continue
        cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", CSRC, src, "-o", out] + MICROBENCH_FLAGS.get(name, [])
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return MICROBENCH_PATH


def build_all(force=False, verbose=False):
    build_library(force, verbose)
    build_host(force, verbose)
    build_microbench(force, verbose)
    return LIB_PATH
