"""The counter files bench.py quotes (roofline.bound / traffic / frac_rocprof) are tied to the binary they describe, and objects built with an
experiment's compiler flags are never reused by a plain build (ADVICE r05 _build.py)."""
import importlib.util
import json
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _tree(tmp_path):
    """A scratch copy of what the state check looks at: the kernel sources and a stamp made from them."""
    from fastecc_amd import _build
    csrc = tmp_path / "csrc"
    csrc.mkdir()
    for name in _build.COUNTED_SOURCES:
        shutil.copy(os.path.join(_build.CSRC, name), csrc / name)
    prof = tmp_path / "profiles" / "rXX"
    prof.mkdir(parents=True)
    stamp = {"sources": _build.kernel_sources_sha256(csrc=str(csrc)), "profile_names": ["tile_dif10_w32", "tile_mid9_w32", "tile_dit10_w32"], "session": "test"}
    (prof / "counters_stamp.json").write_text(json.dumps(stamp))
    return csrc, os.path.join("profiles", "rXX", "counters_stamp.json")


def test_counters_are_quoted_only_for_the_sources_they_were_taken_from(bench, tmp_path):
    csrc, stamp = _tree(tmp_path)
    names = ["tile_dif10_w32", "tile_mid9_w32", "tile_dit10_w32"]
    ok = bench.counters_state(root=str(tmp_path), stamp=stamp, loaded_kernels=names, csrc=str(csrc))
    assert ok["status"] == "ok" and ok["session"] == "test"
    # a kernel source moves on: STALE, and the reason names the file
    with open(csrc / "tile_kernels.hip", "a") as f:
        f.write("// edited after the counters were taken\n")
    st = bench.counters_state(root=str(tmp_path), stamp=stamp, loaded_kernels=names, csrc=str(csrc))
    assert st["status"] == "STALE" and "tile_kernels.hip" in st["why"] and "gf.hpp" not in st["why"]


def test_an_edited_stamp_or_an_unknown_kernel_is_stale_and_no_stamp_is_unstamped(bench, tmp_path):
    csrc, stamp = _tree(tmp_path)
    path = tmp_path / stamp
    rec = json.loads(path.read_text())
    rec["sources"]["sha256"] = "0" * 64
    path.write_text(json.dumps(rec))
    assert bench.counters_state(root=str(tmp_path), stamp=stamp, loaded_kernels=[], csrc=str(csrc))["status"] == "STALE"
    csrc2, stamp2 = _tree(tmp_path / "b") if (tmp_path / "b").mkdir() is None else (None, None)
    st = bench.counters_state(root=str(tmp_path / "b"), stamp=stamp2, loaded_kernels=["tile_mid10_w32"], csrc=str(csrc2))
    assert st["status"] == "STALE" and "tile_mid10_w32" in st["why"]  # another plan's kernel: the files hold no counters for it
    os.remove(path)
    assert bench.counters_state(root=str(tmp_path), stamp=stamp, loaded_kernels=[], csrc=str(csrc))["status"] == "unstamped"


def test_the_committed_stamp_describes_this_tree(bench):
    """The counter files under profiles/<round> were taken from the kernels as they are now (regenerate them with
    tools/sessions/gpu_r06_counters.sh after touching tile_kernels.hip / gf.hpp / ntt_device.hpp / kernels.hpp)."""
    st = bench.counters_state(loaded_kernels=["tile_dif10_w32", "tile_mid9_w32", "tile_dit10_w32"])
    assert st["status"] == "ok", st
    for name in ("rocprofv3_kernel_stats_bench_default.csv", "pmc_valu_default_plan.json", "pmc_traffic.json"):
        assert os.path.exists(os.path.join(ROOT, "profiles", bench.PROFILE_ROUND, name)), name
    assert bench.rocprof_avg_ms("tile_mid9_w32") and bench.pmc_traffic("tile_mid9_w32")


def test_objects_of_another_flag_set_are_not_reused(tmp_path):
    from fastecc_amd import _build
    stamp = tmp_path / ".hip_flags"
    assert _build.flags_changed(str(stamp)) is True  # nobody recorded their flags: rebuild
    stamp.write_text(_build.flags_key() + "\n")
    assert _build.flags_changed(str(stamp)) is False
    ablation = _build.HIP_FLAGS + ["-DFASTECC_DIRECT_ABLATION"]
    assert _build.flags_changed(str(stamp), ablation) is True  # an ablation build after a plain one recompiles ...
    stamp.write_text(_build.flags_key(ablation) + "\n")
    assert _build.flags_changed(str(stamp)) is True  # ... and so does the plain build after it
