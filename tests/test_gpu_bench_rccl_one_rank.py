"""bench.py's N > 1 code against the real RCCL, on ONE GPU.

RCCL refuses two ranks on one device, so the multi-rank job itself runs here only over gloo (tests/test_gpu_bench_multirank.py).  What a 1-GPU
box can do is bring up a ONE-rank `nccl` process group and let bench.py treat it as a multi-rank job (FASTECC_BENCH_TEST_ONE_RANK_GROUP) with
the sharding collectives forced on (FASTECC_SHARDING_FORCE_COLLECTIVES): init_process_group(device_id=...), the device-side MAX / MIN
reductions, a communicator per later mode, the object gather of the device records, the parity gather of the checks and the teardown are
then the very calls of the 8-GPU run, executed by the library that will execute them there."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(extra):
    env = dict(os.environ, FASTECC_BENCH_TEST_ONE_RANK_GROUP="1", FASTECC_SHARDING_FORCE_COLLECTIVES="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    env.pop("FASTECC_BENCH_BACKEND", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-other-paths",
                        "--startup-timeout", "120"] + extra, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and lines, (r.stdout[-1500:], r.stderr[-3000:])
    return json.loads(lines[-1])


def test_bench_distributed_code_on_real_rccl_with_one_rank(hip_lib):
    line = _run(["--log2k", "12"])
    if "UNAVAILABLE" in str(line.get("collectives", "")):  # the library itself did not come up on this box: nothing of bench.py's N > 1 code ran
        pytest.skip("RCCL could not initialise a one-rank group here")
    one = line["one_stripe"]
    assert one.get("complete"), one
    for mode in ("compute_only", "all_to_all", "exchange_only", "all_to_all_in_out", "gather_to_root"):
        assert "ms_per_stripe" in one[mode], (mode, one[mode])
    checks = one["checks"]
    assert checks["slabs_equal_compute_only_on_every_rank"] is True, checks
    assert checks["all_to_all"]["status"] == "ok" and checks["all_to_all_in_out"]["status"] == "ok", checks
    assert line["distributed"]["backend"] == "nccl" and line["distributed"]["rccl"], line["distributed"]
    assert "FAILED" not in str(line["parity_check"]), line["parity_check"]


def test_bench_distributed_code_on_real_rccl_64bit_field(hip_lib):
    line = _run(["--log2k", "11", "--field", "p61", "--block-bytes", "4096"])
    if "UNAVAILABLE" in str(line.get("collectives", "")):
        pytest.skip("RCCL could not initialise a one-rank group here")
    one = line.get("one_stripe")
    if one is None:  # this field's one-stripe modes need several ranks' worth of columns: only the process-group code ran
        assert line["distributed"]["backend"] == "nccl"
        return
    assert one.get("complete"), one
    assert "ms_per_stripe" in one["all_to_all"], one
