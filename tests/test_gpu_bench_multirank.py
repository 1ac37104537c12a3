"""bench.py at N > 1 (BASELINE configs[3]) must not lose its `value` to a failing or stalled secondary mode.

The driver launches `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` over RCCL, one GPU per rank.  A 1-GPU box runs
the same control flow with FASTECC_BENCH_BACKEND=gloo (every rank on device 0, collectives staged through host memory) at a small size;
FASTECC_BENCH_TEST_STALL injects a failure or a stall into one named mode.  `value` is the all_to_all mode, which runs before every other
exchanging mode; each mode has its own timer (--mode-timeout)."""
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


def _bench(world, fault=None, mode_timeout=120, log2k=12, extra=()):
    env = dict(os.environ, FASTECC_BENCH_BACKEND="gloo", OMP_NUM_THREADS="2")
    env.pop("FASTECC_BENCH_TEST_STALL", None)
    if fault:
        env["FASTECC_BENCH_TEST_STALL"] = fault
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1",
           "--log2k", str(log2k), "--mode-timeout", str(mode_timeout), "--sharded-timeout", "400", *extra]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "bench.py must print exactly one JSON line (rc %d)\n%s\n%s" % (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    line = json.loads(lines[0])
    line["_rc"] = r.returncode
    return line


def _value_is_all_to_all(line, world):
    one = line["one_stripe"]
    a2a = one["all_to_all"]
    assert "ms_per_stripe" in a2a, one
    assert line["scaling"] == "strong" and line["n_gpus"] == world and line["value_kind"] == "one_stripe_all_to_all"
    assert line["value"] == a2a["GBps"] and line["ms_per_step"] == a2a["ms_per_stripe"]
    assert line["compute_only_GBps"] == one["compute_only"]["GBps"]  # the exchange-free figure sits beside `value`
    assert line["replicas"]["scaling"] == "weak"
    return one


def test_two_rank_line_all_modes(hip_lib):
    line = _bench(2)
    one = _value_is_all_to_all(line, 2)
    assert one["complete"] is True and "stalled_in" not in one and line["complete"] is True and line["_rc"] == 0
    assert line["expected_shape"].startswith("N=2 is SLOWER") and line["link_peak_GBps"] == one["link_peak_GBps"] and line["link_peak_assumed_GBps"] == 76.8
    assert one["mode_order"][:3] == ["compute_only", "all_to_all", "exchange_only"] and one["mode_order"][-1] == "gather_to_root"
    for name in one["mode_order"]:
        assert "ms_per_stripe" in one[name], (name, one[name])
    # the link figure is the measured one (exchange_only of this run); the assumption is kept beside it
    assert one["link_peak_source"] == "exchange_only of this run" and one["link_peak_GBps"] > 0 and one["link_peak_assumed_GBps"] == 76.8
    assert one["exchange_only"]["link_roofline_frac"] == 1.0
    # (a ratio of two host-staged gloo timings: the bound only catches a nonsensical value — under the sanitizer build the encode side of
    #  all_to_all is not slowed down and exchange_only is, which gave 2.5)
    assert 0 < one["all_to_all"]["link_roofline_frac"] <= 10
    assert one["checks"]["slabs_equal_compute_only_on_every_rank"] is True
    assert one["checks"]["all_to_all"]["status"] == "ok" and one["checks"]["all_to_all"]["equals_gather_to_root"] is True
    assert one["checks"]["all_to_all_in_out"]["status"] == "ok"
    assert line["exchange_only_ms_per_stripe"] == one["exchange_only"]["ms_per_stripe"]


@pytest.mark.parametrize("mode", ["gather_to_root"])
def test_a_failing_mode_costs_only_itself(hip_lib, mode):
    line = _bench(2, fault=mode + ":raise_all")
    one = _value_is_all_to_all(line, 2)
    assert "error" in one[mode] and "injected failure" in one[mode]["error"]
    for name in one["mode_order"]:
        if name != mode:
            assert "ms_per_stripe" in one[name], (name, one[name])
    assert one["complete"] is True
    assert one["checks"]["all_to_all"]["status"] == "ok"


def test_a_stalled_gather_cannot_take_the_value(hip_lib):
    """The last rank never enters gather_to_root's collectives: that mode's timer prints the line, `value` is still all_to_all."""
    line = _bench(2, fault="gather_to_root:stall", mode_timeout=6)
    one = _value_is_all_to_all(line, 2)
    assert one["stalled_in"] == "gather_to_root" and one["complete"] is False
    assert line["complete"] is False and line["_rc"] != 0  # the watchdog's line is marked, and the launcher sees a failure
    assert "error" in one["gather_to_root"] and "stalled" in one["gather_to_root"]["error"]
    for name in ("compute_only", "all_to_all", "exchange_only", "all_to_all_in_out"):
        assert "ms_per_stripe" in one[name]


def test_a_rank_that_never_arrives_leaves_the_replica_line(hip_lib):
    """Nothing of the one-stripe section completes: the line falls back to the replica measurement and says so."""
    line = _bench(2, fault="all:stall", mode_timeout=6)
    assert line["scaling"] == "weak" and "REPLICAS" in line["metric"]
    assert line["one_stripe"]["stalled_in"] in ("setup", "compute_only")  # rank 0 waits for the missing rank in the first agreement
    assert line["value"] == line["replicas"]["value"] > 0
    assert line["value_kind"] == "replicas_fallback" and line["complete"] is False and line["_rc"] != 0


def test_a_process_group_that_never_comes_up_leaves_a_line(hip_lib):
    """One rank never joins torch.distributed: rank 0 prints its own timing (no collective was possible) instead of nothing."""
    line = _bench(2, fault="startup", extra=("--startup-timeout", "8"))
    assert "UNAVAILABLE" in line["collectives"] and line["scaling"] == "weak"
    # only what was measured: one GPU; the two-rank extrapolation is in a key of its own and the run counts as failed
    assert line["n_gpus"] == 1 and line["requested_gpus"] == 2 and line["complete"] is False and line["value_kind"] == "rank0_only" and line["_rc"] != 0
    assert line["value"] > 0 and line["ms_per_step"] == line["rank0_local_ms_per_step"]
    assert abs(line["replicas_extrapolated_GBps"] - 2 * line["value"]) < 0.02


def test_eight_rank_headline_geometry_control_flow(hip_lib):
    """Eight ranks, 4 KB blocks: 128-word slabs in two 64-word sub-slabs, k/8 whole blocks per rank (the geometry of configs[3])."""
    line = _bench(8)
    one = _value_is_all_to_all(line, 8)
    assert one["sub_slabs"] == 2 and one["complete"] is True and line["complete"] is True
    assert "all-to-all" in line["expected_shape"]
    assert one["checks"]["all_to_all"]["status"] == "ok" and one["checks"]["all_to_all_in_out"]["status"] == "ok"


def test_eight_rank_64bit_field_control_flow(hip_lib):
    """BASELINE configs[4]'s line (GF((2^61-1)^2), 64 KB blocks, eight ranks) at a reduced k: the same modes, `value` = all_to_all, and the
    block-distributed parity pinned to the CPU oracle (one element column) as well as to the gathered copy."""
    line = _bench(8, extra=("--field", "p61"))
    one = _value_is_all_to_all(line, 8)
    assert line["dtype"] == "u64" and "2^61" in line["config"]["workload"] and one["complete"] is True and line["complete"] is True
    assert one["checks"]["slabs_equal_compute_only_on_every_rank"] is True
    a2a = one["checks"]["all_to_all"]
    assert a2a["status"] == "ok" and a2a["equals_gather_to_root"] is True and a2a["oracle_column"] == "ok"
    assert "all_to_all_in_out" not in one or "ms_per_stripe" not in (one.get("all_to_all_in_out") or {})  # (needs a stripe every rank can derive: 32-bit field only)
    assert line["parity_check"]["status"] == "ok"
