#!/usr/bin/env python3
"""bench.py — encode GB/s at (n,k) = (2^20, 2^19), 4 KB blocks (BASELINE.json metric), 1..8 MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one fastecc_encode of one stripe (k = 2^19 data blocks of 4 KB -> 2^19 parity blocks), inputs
already resident in HBM, called through the C ABI (include/fastecc.h) exactly as a C++ host would.
Multi-GPU: one process per GPU.  At N > 1 `value` is BASELINE.json configs[3]: ONE stripe sharded over the N GPUs in column
slabs (word columns are independent transforms: no exchange inside the encode), the parity handed over block-distributed by
an RCCL all-to-all (rank g ends with parity blocks [g*M/N, (g+1)*M/N) whole) — scaling "strong".  `one_stripe` carries that
mode next to compute-only (also at the top level as `compute_only_GBps`: the same stripe with no exchange), the bare exchange, the
block-distributed-input form and gather-to-root — in that order, `value` first, each under its own timer, a failing mode costing only
itself — each with the bytes its busiest rank receives and the fraction of the link rate the bare exchange reached in the same run;
`replicas` carries the weak-scaling mode (every rank encodes its own independent stripe, no collective).  A single-process run that sees several GPUs additionally times the
C-ABI form (fastecc_create_sharded: peer copies instead of RCCL) in a child process.

At N = 1 the line additionally carries `other_paths`: short, checked timings of the rows around the headline path (few-loss repair
and a 2 % loss pattern, a code with 4 parity blocks, a mixed-radix order) and BASELINE configs[4] (the 64-bit field, 32 + 32 GiB),
measured in a child process — so that the driver's record holds them too; none of it enters `value` (--no-other-paths skips it).

Throughput convention = the reference's (RS.cpp:38): bytes = data + parity = 2*k*block_bytes per encode,
reported in GB/s (1e9).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import threading
import time

def usable_cpus():
    """CPUs this process may really use: the affinity mask, capped by the cgroup CPU quota (GPU boxes expose 256
    hardware threads but may grant a container far fewer)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except Exception:
        pass
    return n


if int(os.environ.get("WORLD_SIZE", "1")) == 1:
    # CPU baseline (single-process runs only): one OpenMP thread per usable CPU, spinning at barriers (avoids the
    # passive-wait stalls of the reference's many small parallel regions).  Must be set before libgomp
    # loads.  Multi-rank runs must not have hundreds of spinning OpenMP workers per rank on the host.
    os.environ.setdefault("OMP_NUM_THREADS", str(usable_cpus()))
    os.environ.setdefault("OMP_WAIT_POLICY", "active")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

from bench_common import P, P61, random_stripe, random_stripe_p61, splitmix_window, time_steps  # noqa: E402,F401
from bench_multi import (ACTIVE_TEST_HOOKS as _ACTIVE_TEST_HOOKS, EXIT_NO_GROUP, EXIT_WATCHDOG, cabi_sharded_child, dist_on,  # noqa: E402,F401
                         one_stripe_modes)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)
# Secondary (informational) bound: the path is integer-VALU bound once fused.  Chip-wide rate of
# the radix-2 GF(p) butterfly measured in isolation (tools/microbench.hip radix: 3796 radix-2 / 3826-3829 radix-4 form,
# profiles/r03/microbench_radix4_vs_radix2.jsonl).
VALU_PEAK_GBFLY = 3830.0
VALU_PEAK_CLOCK_GHZ = 2.39  # the clock that loop sustains (profiles/r02/microbench_bfly_sustained_r02.jsonl: 2393 MHz at 1055 W; profiles/r05: GRBM cycles / duration)
PROFILE_ROUND = "r06"  # the directory under profiles/ whose counter files this file quotes (regenerated together, see counters_stamp.json)
PMC_VALU = os.path.join("profiles", PROFILE_ROUND, "pmc_valu_default_plan.json")
COUNTERS_STAMP = os.path.join("profiles", PROFILE_ROUND, "counters_stamp.json")
PMC_VALU_P61 = os.path.join("profiles", "r05", "pmc_valu_p61.json")  # the 64-bit field at the configs[4] size (tools/pmc_valu_p61.py)


def counters_state(root=ROOT, stamp=COUNTERS_STAMP, loaded_kernels=None, csrc=None):
    """Do the committed counter files (roofline.bound, .traffic, .frac_rocprof, .valu.issue_floor_frac are quoted from them, not measured in
    this run) describe the binary that is running?  {"status": "ok" | "STALE" | "unstamped", "why": ...}.
      * STALE when the sha256 of the headline kernels' sources in the tree differs from the one recorded when the counters were taken
        (tools/stamp_counters.py), or when a kernel of the loaded library (enc.profile_read() names) is not among the recorded ones;
      * unstamped when there is no stamp file (counter files of unknown origin are not quoted either)."""
    try:
        with open(os.path.join(root, stamp)) as f:
            rec = json.load(f)
    except Exception:
        return {"status": "unstamped", "why": "no %s" % stamp}
    try:
        from fastecc_amd import _build
        now = _build.kernel_sources_sha256(csrc=csrc)
    except Exception as e:  # noqa: BLE001 — no sources beside the library: cannot vouch for the files
        return {"status": "STALE", "why": "the kernel sources cannot be hashed here: %r" % e}
    if now["sha256"] != (rec.get("sources") or {}).get("sha256"):
        changed = sorted(f for f, h in now["files"].items() if (rec.get("sources") or {}).get("files", {}).get(f) != h)
        return {"status": "STALE", "why": "sources changed since the counters were taken: %s" % ", ".join(changed), "stamp": stamp}
    missing = sorted(set(loaded_kernels or ()) - set(rec.get("profile_names") or ()))
    if missing:
        return {"status": "STALE", "why": "kernels of the loaded library without counters: %s" % ", ".join(missing), "stamp": stamp}
    return {"status": "ok", "stamp": stamp, "sources_sha256": now["sha256"], "session": rec.get("session")}


def pmc_valu(path=PMC_VALU):
    """The committed counter summary of the default plan's kernels at the headline size (tools/pmc_valu.py), or None."""
    try:
        with open(os.path.join(ROOT, path)) as f:
            return json.load(f)
    except Exception:
        return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log2k", type=int, default=19, help="k = 2^log2k data blocks (headline: 19)")
    ap.add_argument("--log2m", type=int, default=None, help="parity blocks = 2^log2m (default: = k, the reference's configuration; "
                                                            "k/2 .. k/16 are supported, not the headline metric)")
    ap.add_argument("--log2-n-over-k", type=int, default=1, help="n = 2^e k: e = 2, 3 give 3k, 7k parity blocks (nested cosets)")
    ap.add_argument("--batch", type=int, default=1, help="stripes per step, stored back to back and encoded by fastecc_encode_batch "
                                                         "(small codes; not the headline metric)")
    ap.add_argument("--block-bytes", type=int, default=0, help="default 4096 (65536 with --field p61)")
    ap.add_argument("--field", choices=["fff00001", "p61"], default="fff00001",
                    help="p61 = GF((2^61-1)^2), the 64 KB-block configuration of BASELINE.json configs[4] (not the headline metric)")
    ap.add_argument("--plan", type=int, default=0, help="kernel plan (0 = library default); see DESIGN.md")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-log2k", type=int, default=None, help="sample size for the CPU baseline (default: same as --log2k)")
    ap.add_argument("--no-sharded", action="store_true", help="skip the one_stripe modes")
    ap.add_argument("--sharded-timeout", type=int, default=300,
                    help="N > 1: seconds the one_stripe measurements may take before the line is printed without it (0 = wait forever)")
    ap.add_argument("--startup-timeout", type=int, default=240,
                    help="N > 1: seconds the process group may take to initialise and pass its first barrier; at expiry rank 0 prints a line from its own "
                         "timing (0 = wait forever)")
    ap.add_argument("--mode-timeout", type=int, default=90,
                    help="N > 1: seconds ONE one_stripe mode (warm-up + K steps, incl. the first use of its communicator) may take; at expiry the line "
                         "is printed with every mode measured so far and the job ends (0 = no per-mode timer)")
    ap.add_argument("--sub-slabs", type=int, default=2, help="column sub-slabs of the exchange pipelines (one_stripe)")
    ap.add_argument("--no-parity-check", action="store_true", help="skip the golden-hash gate after the timed region")
    ap.add_argument("--no-other-paths", action="store_true", help="skip the short timings of the widened rows (other_paths)")
    ap.add_argument("--cabi-sharded-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--other-paths-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--slabs", type=int, default=0, help="column slabs on internal streams (0 = library default)")
    ap.add_argument("--option", action="append", default=[], help="library tuning option name=value (fastecc_set_option)")
    return ap.parse_args()


def cpu_baseline_p61(log2k, block_bytes):
    """No reference code exists for this field: the baseline is our plain-C oracle on a bounded sample
    (same k, 1 KiB slices of the blocks: columns are independent, so the rate per byte is the same)."""
    from oracle import Oracle, OracleP61
    N, elems = 1 << log2k, min(block_bytes // 16, 64)
    orc = OracleP61()
    x = orc.fill_splitmix(N, elems, 0x1234)
    t0 = time.perf_counter()
    orc.lib.orc61_encode(x, N, elems)
    dt = time.perf_counter() - t0
    cores = Oracle().num_threads()
    return {"value": round(2.0 * N * elems * 16 / dt / 1e9, 4), "unit": "GB/s", "cores": min(cores, elems), "kind": "port",
            "sample": "one encode of k=2^%d blocks x %d B (a %d-element column slice of the %d B blocks) by oracle/fastecc_oracle_p61.c "
                      "(plain C, 128-bit %% arithmetic, OpenMP over columns), %.2f s" % (log2k, elems * 16, elems, block_bytes, dt)}


def cpu_baseline(log2k, block_bytes):
    """FastECC's own CPU path on this host's cores: the unmodified reference (oracle/_ref, AVX2+OpenMP
    build) when it was prebuilt, else our plain-C port.  One untimed warm-up, then best of 2."""
    from oracle import Oracle, Reference
    N, S = 1 << log2k, block_bytes // 4
    orc = Oracle()
    x = orc.fill_splitmix(N, S, 0x1234)
    if Reference.available(avx2=True):
        ref, kind = Reference(avx2=True), "reference"
        run = lambda buf: ref.encode_inplace(buf)  # noqa: E731
        what = "FastECC RS.cpp:41-63 call sequence on MFA_NTT, g++ -O3 -fopenmp -mavx2 -DSIMD=AVX2"
    else:
        kind = "port"
        run = lambda buf: orc.encode_fast_inplace(buf)  # noqa: E731
        what = "oracle/fastecc_oracle.c (plain C + OpenMP)"
    best = None
    for it in range(3):
        buf = x.copy()
        t0 = time.perf_counter()
        run(buf)
        dt = time.perf_counter() - t0
        if it > 0:
            best = dt if best is None else min(best, dt)
    cores = orc.num_threads()
    return {"value": round(2.0 * N * S * 4 / best / 1e9, 3), "unit": "GB/s", "cores": cores, "kind": kind,
            "sample": "one full encode of k=2^%d blocks x %d B (%s), best of 2 after a warm-up, %.2f s each, %d OpenMP threads"
                      % (log2k, block_bytes, what, best, cores)}


def pmc_traffic(kernel):
    """HBM bytes per launch for `kernel` from a committed rocprofv3 --pmc summary, if there is one."""
    path = os.path.join(ROOT, "profiles", PROFILE_ROUND, "pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(kernel)
    except Exception:
        return None


ROCPROF_STATS = os.path.join("profiles", PROFILE_ROUND, "rocprofv3_kernel_stats_bench_default.csv")


def rocprof_avg_ms(kernel):
    """Average duration of `kernel` (the library's profile name, e.g. tile_mid10_w32) in the committed rocprofv3 --kernel-trace --stats
    summary of this command (captured in the same GPU session as profiles/r04/bench_n1_default.json), or None.
    tile_<mode><levels>_w32[_r16] <-> ntt_tile_kernel<levels, 5 or 4, true, 0/1/2, ...>."""
    import csv
    import re
    m = re.match(r"tile_(dif|dit|mid)(\d+)_w(32|64)(_r16)?$", kernel)
    if not m:
        return None
    want = (int(m.group(2)), 4 if m.group(4) else 5, "true" if m.group(3) == "32" else "false", {"dif": 0, "dit": 1, "mid": 2}[m.group(1)])
    try:
        with open(os.path.join(ROOT, ROCPROF_STATS)) as f:
            for row in csv.DictReader(f):
                t = re.search(r"ntt_tile_kernel<(\d+), (\d+), (true|false), (\d+)", row.get("Name", ""))
                if t and (int(t.group(1)), int(t.group(2)), t.group(3), int(t.group(4))) == want:
                    return float(row["AverageNs"]) / 1e6
    except Exception:
        return None
    return None


def parity_check(enc, log2k, block_bytes, device):
    """The gate BASELINE.md §3.4 specifies: encode the splitmix(0x1234) stripe with the context that was just timed and
    compare the parity hash (main.cpp:202-212) with the value recorded from the unmodified reference.  The oracle
    only generates the input and hashes the output here (checker, outside every timed region)."""
    import numpy as np
    from oracle import Oracle
    with open(os.path.join(ROOT, "tests", "golden", "golden_hashes.json")) as f:
        gold = json.load(f)
    N, S = 1 << log2k, block_bytes // 4
    want = None
    for c in gold.get("survey_appendix_b", []) + gold.get("cases", []):
        if c.get("input") == "splitmix" and c.get("log2N") == log2k and c.get("block_bytes") == block_bytes:
            want = c
            break
    if want is None:
        return {"status": "skipped", "why": "no golden hash for k=2^%d, %d-byte blocks" % (log2k, block_bytes)}
    orc = Oracle()
    x = orc.fill_splitmix(N, S, gold["splitmix_seed"])
    if orc.hash(x) != want["hash_input"]:
        return {"status": "FAILED", "why": "input generator does not reproduce the recorded input hash"}
    d = torch.from_numpy(x.view(np.int32)).to(device)
    out = torch.empty_like(d)
    enc.encode(d, out, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = orc.hash(out.cpu().numpy().view(np.uint32).reshape(N, S))
    return {"status": "ok" if got == want["hash_parity"] else "FAILED", "hash": got, "expected": want["hash_parity"],
            "what": "parity hash (main.cpp:202-212) of the splitmix64(0x%x) stripe vs the unmodified reference's" % gold["splitmix_seed"]}


def parity_check_p61(data, parity, k, block_bytes):
    """No reference output exists for this field (parity unpinned upstream): the gate re-encodes two element columns of
    the stripe that was just timed with this repository's oracle (checker only) and compares them with the parity."""
    import numpy as np
    from oracle import OracleP61
    elems = block_bytes // 16
    cols = [0, elems - 1]
    idx = torch.tensor([2 * c + j for c in cols for j in (0, 1)], device=data.device)
    x = data.view(k, 2 * elems)[:, idx].cpu().numpy().view(np.uint64)
    got = parity.view(k, 2 * elems)[:, idx].cpu().numpy().view(np.uint64)
    want = OracleP61().encode(np.ascontiguousarray(x))
    return {"status": "ok" if np.array_equal(got, want) else "FAILED",
            "what": "element columns %s of the timed stripe re-encoded by oracle/fastecc_oracle_p61.c (no upstream output exists for this field)" % cols}


def oracle_columns(data, got, k, S, rows_out, which, cols=(0, -1)):
    """Two word columns of a result re-computed by the CPU oracle (checker only; columns are independent transforms): `which` maps the oracle's
    encoder to the expected rows.  Returns "ok" / "FAILED" — the entry is then pinned to the oracle, not only to another plan of this library."""
    import numpy as np
    from oracle import Oracle
    orc = Oracle()
    idx = torch.tensor([c % S for c in cols], device=data.device)
    x = np.ascontiguousarray(data.view(k, S)[:, idx].cpu().numpy().view(np.uint32))
    want = which(orc, x)
    have = got.view(rows_out, S)[:, idx].cpu().numpy().view(np.uint32)
    return "ok" if want.shape == have.shape and np.array_equal(want, have) else "FAILED"


def other_paths(fastecc_amd, enc, data, parity, log2k, block_bytes, device, stream):
    """Short, checked timings of the rows around the headline path, so that the driver's record carries them too —
    they are NOT the metric.  `data` / `parity` are the codeword the timed context just produced.  Every entry verifies what it timed."""
    import numpy as np
    k, S = 1 << log2k, block_bytes // 4
    out = {"what": "HIP-event timings of other paths at this size, each checked; not part of `value`"}

    def event_ms(fn, reps):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    def single_call_ms(fn, idle_s=0.3, tries=3):
        """One call on an idle device (best of `tries`): back-to-back launches of the matrix-core pass pull the board to its power cap and the shader
        clock drops from 1.95 to 1.1-1.3 GHz after two launches (profiles/r06/direct_pass_clock.json), so the average of a loop is the SUSTAINED figure;
        a repair in the field is one call."""
        best = None
        for _ in range(tries):
            torch.cuda.synchronize()
            time.sleep(idle_s)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1)
            best = t if best is None else min(best, t)
        return best

    # --- erasure decoding on the context that was timed: lost blocks are overwritten, repaired in place and compared with saved copies
    try:
        rng = np.random.default_rng(7)
        dv, pv = data.view(k, S), parity.view(k, S)
        for name, lost in (("repair_1_data_1_parity_lost", np.array([k // 3, k + k // 7])),
                           ("repair_8_data_8_parity_lost", np.r_[rng.permutation(k)[:8], k + rng.permutation(k)[:8]]),
                           ("repair_16_data_lost", rng.permutation(k)[:16]),
                           ("repair_64_data_lost", rng.permutation(k)[:64]),
                           ("repair_128_data_128_parity_lost", np.r_[rng.permutation(k)[:128], k + rng.permutation(k)[:128]]),
                           ("repair_2_percent_of_the_codeword_lost", rng.permutation(2 * k)[: (2 * k) // 50])):
            dp, pp = np.ones(k, np.uint8), np.ones(k, np.uint8)
            dp[lost[lost < k]] = 0
            pp[lost[lost >= k] - k] = 0
            di = torch.from_numpy(np.flatnonzero(dp == 0)).to(device)
            pi = torch.from_numpy(np.flatnonzero(pp == 0)).to(device)
            saved_d, saved_p = dv[di].clone(), pv[pi].clone()
            t0 = time.perf_counter()
            enc.decode_prepare(dp, pp)   # the first set-up of this kind of pattern on the context also builds that path's tables / contexts
            prep_first = (time.perf_counter() - t0) * 1e3
            t0 = time.perf_counter()
            for _ in range(3):
                enc.decode_prepare(dp, pp)
            prep = (time.perf_counter() - t0) / 3 * 1e3

            def once():
                dv[di] = -1
                pv[pi] = -2
                enc.repair(data, parity, stream=stream)

            once()
            ok = bool(torch.equal(dv[di], saved_d)) and bool(torch.equal(pv[pi], saved_p))
            ms = event_ms(lambda: enc.repair(data, parity, stream=stream), 5)
            out[name] = {"prepare_first_ms": round(prep_first, 2), "prepare_steady_ms": round(prep, 2), "repair_ms": round(ms, 3), "restored": ok}
            if name in ("repair_64_data_lost", "repair_128_data_128_parity_lost"):  # the matrix-core pass: sustained (above) and one call on an idle device
                out[name]["repair_single_call_ms"] = round(single_call_ms(lambda: enc.repair(data, parity, stream=stream)), 3)
            if name == "repair_2_percent_of_the_codeword_lost":
                # fastecc_decode alone (the lost data blocks, not the lost parity): the split transform of the (2k,k) layout
                dv[di] = -1
                enc.decode(data, parity, stream=stream)
                ok_d = bool(torch.equal(dv[di], saved_d))
                ms_d = event_ms(lambda: enc.decode(data, parity, stream=stream), 5)
                enc_ms = event_ms(lambda: enc.encode(data, parity, stream=stream), 5)  # (the parity it writes is the parity that is there)
                out["decode_2_percent_of_the_codeword_lost"] = {"decode_ms": round(ms_d, 3), "restored": ok_d, "encode_ms_same_context": round(enc_ms, 3),
                                                                "decode_over_encode": round(ms_d / enc_ms, 3),
                                                                "what": "structurally an encode plus a bit: the data half runs the encoder's three passes (per-block "
                                                                        "factors on the way in, the parity half's k / 32-row transform added inside MID, only the "
                                                                        "rebuilt blocks stored) — price it against the encode, not against the codeword's bytes"}
    except Exception as e:  # noqa: BLE001
        out["decode_error"] = repr(e)
    # --- a code with 4 parity blocks: direct evaluation against the transform pipeline of the same context
    try:
        with fastecc_amd.Encoder(k + 4, k, block_bytes, device=device.index or 0) as small:
            p_direct = torch.empty(4 * S, dtype=torch.int32, device=device)
            p_pipe = torch.empty_like(p_direct)
            ms = event_ms(lambda: small.encode(data, p_direct, stream=stream), 10)
            small.set_option("encode_direct_max", 0)
            small.encode(data, p_pipe, stream=stream)
            stride4 = 1 << min(log2k - 2, 4)  # parity block j = block j * 2^fold of the (2k,k) parity (include/fastecc.h)
            out["encode_k_plus_4_parity"] = {"ms": round(ms, 3), "data_GBps": round(k * block_bytes / ms / 1e6, 1),
                                             "same_parity_as_the_transform_pipeline": bool(torch.equal(p_direct, p_pipe)),
                                             "oracle_columns": oracle_columns(data, p_direct, k, S, 4, lambda orc, x: orc.encode_fast(x)[::stride4][:4])}
    except Exception as e:  # noqa: BLE001
        out["small_m_error"] = repr(e)
    # --- more parity blocks, still one read of the data (matrix cores)
    try:
        with fastecc_amd.Encoder(k + 64, k, block_bytes, device=device.index or 0) as small:
            p_direct = torch.empty(64 * S, dtype=torch.int32, device=device)
            p_pipe = torch.empty_like(p_direct)
            ms = event_ms(lambda: small.encode(data, p_direct, stream=stream), 10)
            small.set_option("encode_direct_max", 0)
            small.encode(data, p_pipe, stream=stream)
            ms_pipe = event_ms(lambda: small.encode(data, p_pipe, stream=stream), 5)
            stride64 = 1 << min(log2k - 6, 4)
            small.set_option("encode_direct_max", 160)
            ms_single = single_call_ms(lambda: small.encode(data, p_direct, stream=stream))
            out["encode_k_plus_64_parity"] = {"ms": round(ms, 3), "data_GBps": round(k * block_bytes / ms / 1e6, 1), "transform_pipeline_ms": round(ms_pipe, 3),
                                              "single_call_ms": round(ms_single, 3),
                                              "ms_is": "the average of 10 back-to-back calls = sustained, with the shader clock pulled down to 1.1-1.3 GHz by the power cap; "
                                                       "single_call_ms = one call on an idle device at 1.95 GHz (profiles/r06/direct_pass_clock.json)",
                                              "same_parity_as_the_transform_pipeline": bool(torch.equal(p_direct, p_pipe)),
                                              "oracle_columns": oracle_columns(data, p_direct, k, S, 64, lambda orc, x: orc.encode_fast(x)[::stride64][:64])}
    except Exception as e:  # noqa: BLE001
        out["k_plus_64_error"] = repr(e)
    # --- the stripe in HOST memory (what RS.cpp times, RS.cpp:25-38): pinned buffers in and out over the host link, column slabs
    #     uploaded, encoded and downloaded in a pipeline (FASTECC_MEM_HOST_PINNED); GB/s counts data + parity bytes like `value`
    try:
        hx = torch.empty(k * S, dtype=torch.int32).pin_memory()
        hp = torch.empty(k * S, dtype=torch.int32).pin_memory()
        hx.copy_(data.cpu())
        ref = parity.clone()
        enc.encode(data, ref, stream=stream)

        def host_once():
            enc.encode(hx, hp, stream=stream, mem=fastecc_amd.MEM_HOST_PINNED)
            torch.cuda.synchronize()

        host_once()
        ok = bool(torch.equal(hp, ref.cpu()))
        t0 = time.perf_counter()
        for _ in range(3):
            host_once()
        hms = (time.perf_counter() - t0) / 3 * 1e3
        # what this box's host link allows, measured in the same process on the same pinned buffers: 2 GiB up, 2 GiB down, and both at once
        # (two streams) — the end-to-end call cannot beat the duplex time
        s_up, s_dn = torch.cuda.Stream(), torch.cuda.Stream()
        scratch = torch.empty_like(data)

        def wall(fn, reps=3):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / reps * 1e3

        # (the copies the library itself issues — hipMemcpy2DAsync, here with whole 4 KiB rows — through the runtime torch has loaded.  Both torch's
        #  copy_ and plain hipMemcpyAsync of the two directions on two streams came out serialised on these boxes, "both at once" no faster
        #  than one direction, while the pitched copies overlap: 57 GB/s against 97 GB/s on the box of profiles/r04/host_link_and_pipeline.jsonl.)
        import ctypes
        hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
        hip.hipMemcpy2DAsync.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]

        def up():
            hip.hipMemcpy2DAsync(scratch.data_ptr(), block_bytes, hx.data_ptr(), block_bytes, block_bytes, k, 1, s_up.cuda_stream)  # hipMemcpyHostToDevice

        def down():
            hip.hipMemcpy2DAsync(hp.data_ptr(), block_bytes, ref.data_ptr(), block_bytes, block_bytes, k, 2, s_dn.cuda_stream)      # hipMemcpyDeviceToHost

        nbytes = float(k * block_bytes)
        up_ms, dn_ms = wall(up), wall(down)
        both_ms = wall(lambda: (up(), down()))
        probe = {"h2d_GBps": round(nbytes / up_ms / 1e6, 1), "d2h_GBps": round(nbytes / dn_ms / 1e6, 1),
                 "duplex_GBps": round(2 * nbytes / both_ms / 1e6, 1), "duplex_ms": round(both_ms, 2),
                 "what": "2 GiB pinned copies on this box in this process (hipMemcpy2DAsync, whole rows): up, down, and both directions at once on two streams (wall clock, 3 each)"}
        out["host_pinned_end_to_end"] = {"ms": round(hms, 2), "GBps": round(2.0 * k * block_bytes / hms / 1e6, 1), "same_parity_as_the_device_encode": ok,
                                         "link_probe": probe, "frac_of_duplex": round(both_ms / hms, 3),
                                         "what": "fastecc_encode(FASTECC_MEM_HOST_PINNED): 2 GiB of data up and 2 GiB of parity down over the host link "
                                                 "inside the timed region (wall clock, 3 calls); the regime RS.cpp:25-38 measures.  frac_of_duplex = "
                                                 "the time both copies alone need side by side / the time of the call"}
        del scratch
        # the drop-in calls themselves, on PAGEABLE memory as RS.cpp:25-33 holds it (malloc'ed buffers; numpy arrays here): the contiguous stripe
        # (fastecc_encode, FASTECC_MEM_HOST) and the reference's T** table of block pointers, in place (fastecc_encode_blocks)
        import numpy as np
        px = hx.numpy().copy()
        pp = np.empty_like(px)
        want = ref.cpu().numpy()

        def wall_host(fn, reps=3):
            fn()
            each = []
            for _ in range(reps):
                t0 = time.perf_counter()
                fn()
                each.append((time.perf_counter() - t0) * 1e3)
            return each

        each = wall_host(lambda: enc.encode_host(px.reshape(k, S), pp.reshape(k, S)))
        pms = sum(each) / len(each)
        hw = os.cpu_count() or 1
        out["host_pageable_end_to_end"] = {"ms": round(pms, 2), "GBps": round(2.0 * k * block_bytes / pms / 1e6, 1), "same_parity_as_the_device_encode": bool(np.array_equal(pp, want)),
                                           "ms_each": [round(t, 2) for t in each], "ms_best": round(min(each), 2),
                                           # what the time depends on besides the link: the helper threads that move 2 + 2 GiB between pageable memory and the
                                           # pinned slots share this container's CPU quota with everything else on the box
                                           "stage_threads": min(6, max(2, hw // 4)), "hardware_threads": hw, "usable_cpus": usable_cpus(),
                                           "omp_threads_env": os.environ.get("OMP_NUM_THREADS"),
                                           "what": "fastecc_encode(FASTECC_MEM_HOST) on pageable buffers, synchronous: upload, encode, parity back through a ring of pinned "
                                                   "slots emptied by helper threads (wall clock, mean of 3 calls after a warm-up; ms_each: the three; the pool's boxes "
                                                   "ran this at 81-97 ms: host-bound, the copying threads of a 16-CPU quota move 4 GiB through the cores)"}
        table = (ctypes.c_void_p * k)(*(px.ctypes.data + np.arange(k, dtype=np.uint64) * np.uint64(block_bytes)).tolist())
        keep = px.copy()

        def blocks_once():
            np.copyto(px, keep)  # (in place: the data is gone after a call; the refill is not timed below)

        times = []
        for _ in range(3):
            blocks_once()
            t0 = time.perf_counter()
            rc = fastecc_amd.lib().fastecc_encode_blocks(enc._h, table)
            times.append((time.perf_counter() - t0) * 1e3)
            if rc != 0:
                raise RuntimeError("fastecc_encode_blocks rc=%d" % rc)
        bms = min(times[1:])
        out["encode_blocks_end_to_end"] = {"ms": round(bms, 2), "GBps": round(2.0 * k * block_bytes / bms / 1e6, 1), "same_parity_as_the_device_encode": bool(np.array_equal(px, want)),
                                           "what": "fastecc_encode_blocks: the reference's T** form (RS.cpp:31-33), k pointers to 4 KiB blocks in pageable memory, in place; "
                                                   "blocks gathered / scattered through the staging rings (wall clock, best of 2 after a first call)"}
        del px, pp, keep, want, table
        del hx, hp, ref
    except Exception as e:  # noqa: BLE001
        out["host_pinned_error"] = repr(e)
    # --- a mixed-radix order (3 * 2^(log2k - 2) data blocks), fused odd-radix tiles against the unfused plan of the same context
    try:
        km = 3 << max(log2k - 2, 1)
        with fastecc_amd.Encoder(2 * km, km, block_bytes, device=device.index or 0, flags=fastecc_amd.CODE_MIXED_RADIX) as mixed:
            src = data[: km * S]
            p_fused, p_plain = torch.empty(km * S, dtype=torch.int32, device=device), torch.empty(km * S, dtype=torch.int32, device=device)
            plan = mixed.plan()
            ms = event_ms(lambda: mixed.encode(src, p_fused, stream=stream), 10)
            mixed.set_option("fuse_radix", 0)
            mixed.encode(src, p_plain, stream=stream)
            out["encode_mixed_radix_3x2^%d" % max(log2k - 2, 1)] = {"ms": round(ms, 3), "GBps": round(2.0 * km * block_bytes / ms / 1e6, 1), "plan": plan,
                                                                     "same_parity_as_the_unfused_plan": bool(torch.equal(p_fused, p_plain)),
                                                                     "oracle_columns": oracle_columns(src, p_fused, km, S, km, lambda orc, x: orc.encode_mixed(x))}
    except Exception as e:  # noqa: BLE001
        out["mixed_radix_error"] = repr(e)
    # --- a composite odd factor of the prime-factor map (21 * 2^(log2k - 5) data blocks: the order the seven plain factors would round up to is 14 % larger)
    try:
        kp = 21 << max(log2k - 5, 1)
        with fastecc_amd.Encoder(2 * kp, kp, block_bytes, device=device.index or 0, flags=fastecc_amd.CODE_MIXED_RADIX_PFA) as pfa:
            src = data[: kp * S]
            p_fused, p_plain = torch.empty(kp * S, dtype=torch.int32, device=device), torch.empty(kp * S, dtype=torch.int32, device=device)
            plan = pfa.plan()
            ms = event_ms(lambda: pfa.encode(src, p_fused, stream=stream), 10)
            pfa.set_option("fuse_radix", 0)
            pfa.encode(src, p_plain, stream=stream)
            out["encode_mixed_radix_21x2^%d" % max(log2k - 5, 1)] = {"ms": round(ms, 3), "GBps": round(2.0 * kp * block_bytes / ms / 1e6, 1), "plan": plan,
                                                                      "same_parity_as_the_unfused_plan": bool(torch.equal(p_fused, p_plain)),
                                                                      "oracle_columns": oracle_columns(src, p_fused, kp, S, kp, lambda orc, x: orc.encode_mixed(x))}
    except Exception as e:  # noqa: BLE001
        out["mixed_radix_pfa_error"] = repr(e)
    return out


def other_field_p61(fastecc_amd, device, stream, steps=10):
    """BASELINE.json configs[4] in the same run: (2^20, 2^19) x 64 KB over GF((2^61-1)^2), 32 + 32 GiB, HBM-resident.
    Timed with HIP events on the launch stream over `steps` encodes (the wall clock around a host-side synchronise is reported next to it:
    rounds 2-3 timed 3 steps that way and every run came out as 200.0 ms / 3), then `steps` more with the per-kernel events on."""
    k, bb = 1 << 19, 65536
    free, _ = torch.cuda.mem_get_info(device)
    if free < 70 * 2**30:
        return {"skipped": "needs 64 GiB of free HBM, %.0f GiB free" % (free / 2**30)}
    data = random_stripe_p61(k * (bb // 8), device, seed=0x61)
    parity = torch.empty_like(data)
    with fastecc_amd.Encoder(2 * k, k, bb, device=device.index or 0, field=fastecc_amd.FIELD_GF_P61_SQUARED) as enc:
        for _ in range(2):
            enc.encode(data, parity, stream=stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(steps):
            enc.encode(data, parity, stream=stream)
        e1.record()
        torch.cuda.synchronize()
        wall_ms = (time.perf_counter() - t0) / steps * 1e3
        ms = e0.elapsed_time(e1) / steps
        enc.profile(True)
        enc.profile_reset()
        for _ in range(steps):
            enc.encode(data, parity, stream=stream)
        kernels = enc.profile_read()
        enc.profile(False)
        per_kernel = {kn: {"avg_ms": round(v[0] / v[1], 3), "alg_GB_per_launch": round(v[2] / v[1] / 1e9, 2),
                           "TBps": round(v[2] / v[0] / 1e9, 3), "frac_of_hbm_peak": round(v[2] / v[0] / 1e6 / HBM_PEAK_GBPS, 4)}
                      for kn, v in sorted(kernels.items())}
        dom = max(kernels.items(), key=lambda kv: kv[1][0]) if kernels else None
        roof = None
        if dom:
            name, (ms_total, launches, nbytes) = dom
            ach = nbytes / ms_total / 1e6
            pmc = pmc_valu(PMC_VALU_P61)  # counters of this size's kernels (tools/pmc_valu_p61.py): DIF7 / DIT7 run nearer a copy's rate, MID5 nearer the issue rate
            ev = (pmc or {}).get("kernels", {}).get(name)
            roof = {"bound": ev["bound"] if ev else "hbm",
                    "bound_source": PMC_VALU_P61 + " (rocprofv3 --pmc, committed; not measured in this run)" if ev else "assumed (no counter file)",
                    "per_kernel_bound": None if not pmc else {kn: {k2: e[k2] for k2 in ("bound", "valu_issue_frac", "hbm_frac_achievable", "clock_GHz")}
                                                              for kn, e in pmc["kernels"].items()},
                    "encode_floors": None if not pmc else {k2: pmc["encode"][k2] for k2 in ("valu_floor_frac", "hbm_floor_frac_achievable", "clock_GHz")},
                    "kernel": name, "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4),
                    "avg_kernel_ms": round(ms_total / launches, 3), "alg_bytes_per_launch": nbytes / launches,
                    "encode": {"achieved": round(2.0 * k * bb / ms / 1e6, 1), "frac": round(2.0 * k * bb / ms / 1e6 / HBM_PEAK_GBPS, 4),
                               "hbm_trips": int(round(sum(v[1] for v in kernels.values()) / steps))}}
        check = parity_check_p61(data, parity, k, bb)
        # the erasure decoder on the same codeword: 2 % of the DATA blocks lost (the even / odd split: an encode plus a transform of k / 32 rows);
        # what was lost is overwritten, decoded in place and compared with saved copies
        decode = None
        try:
            import numpy as np
            rng = np.random.default_rng(61)
            dp, pp = np.ones(k, np.uint8), np.ones(k, np.uint8)
            dp[rng.permutation(k)[: k // 50]] = 0
            di = torch.from_numpy(np.flatnonzero(dp == 0)).to(device)
            dv = data.view(k, -1)
            saved = dv[di].clone()
            t0 = time.perf_counter()
            enc.decode_prepare(dp, pp)
            prep_ms = (time.perf_counter() - t0) * 1e3
            dv[di] = -1
            enc.decode(data, parity, stream=stream)
            ok = bool(torch.equal(dv[di], saved))
            d0, d1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            d0.record()
            for _ in range(3):
                enc.decode(data, parity, stream=stream)
            d1.record()
            torch.cuda.synchronize()
            dms = d0.elapsed_time(d1) / 3
            decode = {"lost_data_blocks": int(k // 50), "prepare_first_ms": round(prep_ms, 1), "decode_ms": round(dms, 3), "restored": ok,
                      "decode_over_encode": round(dms / ms, 3), "what": "the (2k,k) decoder's even / odd split: the data chain is the encoder's five passes with the "
                      "parity half's k / 32-row transform added between the halves of MID"}
            del saved
        except Exception as e:  # noqa: BLE001
            decode = {"error": repr(e)}
        # few losses: the direct path (interpolation on k nodes: the data rows and as many parity blocks as data blocks are lost)
        few = None
        try:
            dp, pp = np.ones(k, np.uint8), np.ones(k, np.uint8)
            dp[[7, k // 2]] = 0
            pp[[11]] = 0
            di = torch.from_numpy(np.flatnonzero(dp == 0)).to(device)
            pi = torch.from_numpy(np.flatnonzero(pp == 0)).to(device)
            dv, pv = data.view(k, -1), parity.view(k, -1)
            saved, saved_p = dv[di].clone(), pv[pi].clone()
            t0 = time.perf_counter()
            enc.decode_prepare(dp, pp)
            prep_ms = (time.perf_counter() - t0) * 1e3
            dv[di] = -1
            pv[pi] = -2
            enc.repair(data, parity, stream=stream)
            ok = bool(torch.equal(dv[di], saved)) and bool(torch.equal(pv[pi], saved_p))
            d0, d1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            d0.record()
            for _ in range(3):
                enc.repair(data, parity, stream=stream)
            d1.record()
            torch.cuda.synchronize()
            few = {"lost": "2 data blocks, 1 parity block", "prepare_first_ms": round(prep_ms, 1), "repair_ms": round(d0.elapsed_time(d1) / 3, 3), "restored": ok,
                   "what": "direct path: one read of the data stripe and 2 parity blocks (32 GiB), three outputs"}
            del saved, saved_p
        except Exception as e:  # noqa: BLE001
            few = {"error": repr(e)}
        return {"workload": "RS encode k=2^19 -> 2^19 parity blocks, 65536 B blocks, GF((2^61-1)^2), 32 GiB stripe", "decode_2_percent_of_the_data_lost": decode,
                "repair_few_lost": few,
                "ms_per_step": round(ms, 3),
                "GBps": round(2.0 * k * bb / ms / 1e6, 1), "steps": steps, "timing": "HIP events on the launch stream around %d encodes" % steps,
                "wall_clock_ms_per_step": round(wall_ms, 3), "plan": enc.plan(), "per_kernel": per_kernel, "roofline": roof, "parity_check": check,
                "parity_pin": "no upstream code exists for this field: pinned to this repository's oracle and Python big-integer goldens"}


def other_field_p61_cosets(fastecc_amd, device, stream, steps=5):
    """n = 4k over the 64-bit field (more parity than data blocks, native transforms): (2^19, 2^17) x 64 KB, 8 GiB of data -> 24 GiB of parity, HIP
    events; two element columns of the result re-computed by the oracle's composition (checker only)."""
    import numpy as np
    from oracle import OracleP61
    k, bb, e = 1 << 17, 65536, 2
    free, _ = torch.cuda.mem_get_info(device)
    if free < 48 * 2**30:
        return {"skipped": "needs 40 GiB of free HBM, %.0f GiB free" % (free / 2**30)}
    data = random_stripe_p61(k * (bb // 8), device, seed=0x614)
    parity = torch.empty(3 * data.numel(), dtype=data.dtype, device=device)
    with fastecc_amd.Encoder(k << e, k, bb, device=device.index or 0, field=fastecc_amd.FIELD_GF_P61_SQUARED) as enc:
        for _ in range(2):
            enc.encode(data, parity, stream=stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(steps):
            enc.encode(data, parity, stream=stream)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / steps
        enc.profile(True)
        enc.profile_reset()
        enc.encode(data, parity, stream=stream)
        kernels = enc.profile_read()
        enc.profile(False)
        plan = enc.plan()
        # the decoder of these codes (the same scheme on the 4k-th roots of unity): 2 % of the DATA blocks lost, decoded in place, compared with saved copies
        decode = None
        try:
            rng = np.random.default_rng(614)
            dp, pp = np.ones(k, np.uint8), np.ones(3 * k, np.uint8)
            dp[rng.permutation(k)[: k // 50]] = 0
            di = torch.from_numpy(np.flatnonzero(dp == 0)).to(device)
            dv = data.view(k, -1)
            saved = dv[di].clone()
            enc.decode_prepare(dp, pp)
            dv[di] = -1
            enc.decode(data, parity, stream=stream)
            ok = bool(torch.equal(dv[di], saved))
            d0, d1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            d0.record()
            for _ in range(2):
                enc.decode(data, parity, stream=stream)
            d1.record()
            torch.cuda.synchronize()
            decode = {"lost_data_blocks": int(k // 50), "decode_ms": round(d0.elapsed_time(d1) / 2, 3), "restored": ok,
                      "what": "x p'(x) at the k data positions only: the way down on all 4k positions (2^19 blocks), one folding MID tile (every fourth position), the way up on k positions; gather through the position map and scatter where the plan's end passes are tiles"}
            del saved
        except Exception as e2:  # noqa: BLE001
            decode = {"error": repr(e2)}
        # few losses: the (2k,k) code of the data and the first coset decodes them by its direct path (a read of 2k blocks, no transform over n)
        few = None
        try:
            dp, pp = np.ones(k, np.uint8), np.ones(3 * k, np.uint8)
            dp[[5, k // 3, k - 1]] = 0
            pp[[7, k + 11]] = 0
            di = torch.from_numpy(np.flatnonzero(dp == 0)).to(device)
            dv = data.view(k, -1)
            saved = dv[di].clone()
            enc.decode_prepare(dp, pp)
            dv[di] = -1
            enc.decode(data, parity, stream=stream)
            ok = bool(torch.equal(dv[di], saved))
            d0, d1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            d0.record()
            for _ in range(3):
                enc.decode(data, parity, stream=stream)
            d1.record()
            torch.cuda.synchronize()
            few = {"lost": "3 data blocks, 1 block of the first coset, 1 of the second", "decode_ms": round(d0.elapsed_time(d1) / 3, 3), "restored": ok,
                   "what": "direct path of the (2k,k) code inside (data + first coset): one read of the data stripe and 3 parity blocks of the first coset (8 GiB)"}
            del saved
        except Exception as e2:  # noqa: BLE001
            few = {"error": repr(e2)}
    o = OracleP61()
    elems = bb // 16
    cols = [0, elems - 1]
    idx = torch.tensor([2 * c + j for c in cols for j in (0, 1)], device=device)
    x = data.view(k, 2 * elems)[:, idx].cpu().numpy().view(np.uint64)
    got = parity.view(3 * k, 2 * elems)[:, idx].cpu().numpy().view(np.uint64)
    gens = []
    for j in range(1, e + 1):
        w = o.root(k << j)
        gens += [o.cpow(w, c) for c in range(1, 1 << j, 2)]
    coef = o.ntt(np.ascontiguousarray(x), inverse=True)
    inv_n = o.cinv((k % P61, 0))
    want = np.concatenate([o.ntt(o.scale_blocks(coef, inv_n, g)) for g in gens])
    return {"workload": "RS encode k=2^17 data -> 3 x 2^17 parity blocks (n = 4k), 65536 B blocks, GF((2^61-1)^2): 8 GiB in, 24 GiB out", "ms_per_step": round(ms, 3),
            "GBps_data_plus_parity": round(4.0 * k * bb / ms / 1e6, 1), "steps": steps, "plan": plan, "decode_2_percent_of_the_data_lost": decode, "decode_few_lost": few,
            "launches_per_encode": {kn: v[1] for kn, v in sorted(kernels.items())}, "per_kernel_avg_ms": {kn: round(v[0] / v[1], 3) for kn, v in sorted(kernels.items())},
            "parity_check": {"status": "ok" if np.array_equal(got, want) else "FAILED",
                             "what": "element columns %s of all three cosets re-computed by the oracle's composition (iNTT, block i *= g^i / N, NTT per coset generator)" % cols},
            "parity_pin": "no upstream code exists for this field: pinned to this repository's oracle and Python big-integer goldens (tests/golden/golden_p61.json coset_cases)"}


def other_paths_child(args):
    """The widened rows in a process of their own (one JSON object, progressively: the parent keeps the last complete line)."""
    import fastecc_amd
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    k, bb = 1 << args.log2k, args.block_bytes or 4096
    stream = torch.cuda.current_stream().cuda_stream
    data = random_stripe(k * (bb // 4), device, seed=0x1234)
    parity = torch.empty_like(data)
    out = {}
    with fastecc_amd.Encoder(2 * k, k, bb, device=0) as enc:
        enc.encode(data, parity, stream=stream)
        torch.cuda.synchronize()
        try:
            out = other_paths(fastecc_amd, enc, data, parity, args.log2k, bb, device, stream)
        except Exception as e:  # noqa: BLE001
            out = {"error": repr(e)}
    print(json.dumps(out), flush=True)
    del data, parity
    torch.cuda.empty_cache()
    if args.log2k == 19:
        try:
            out["configs4_64bit_field"] = other_field_p61(fastecc_amd, device, stream)
        except Exception as e:  # noqa: BLE001
            out["configs4_64bit_field"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
        torch.cuda.empty_cache()
        try:
            out["p61_n_equals_4k"] = other_field_p61_cosets(fastecc_amd, device, stream)
        except Exception as e:  # noqa: BLE001
            out["p61_n_equals_4k"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)


#                                         the line prices the modes against the rate `exchange_only` measures in the same run and keeps this as link_peak_assumed


def main():
    args = parse()
    if args.cabi_sharded_child:
        return cabi_sharded_child(args)
    if args.other_paths_child:
        return other_paths_child(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if world == 1 and args.gpus > 1:
        raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product has no CPU path)"
    # Test hook (1-GPU boxes): FASTECC_BENCH_BACKEND=gloo runs the multi-rank control flow with every rank on device 0 and
    # the collectives staged through host memory.  The driver's runs use RCCL ("nccl"), one GPU per rank.
    backend = os.environ.get("FASTECC_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    from fastecc_amd import _build
    if not os.path.exists(_build.LIB_PATH):
        _build.build_library()
    import fastecc_amd

    k = 1 << args.log2k
    m_blocks = k if args.log2m is None else 1 << args.log2m
    if args.log2_n_over_k > 1:
        m_blocks = ((1 << args.log2_n_over_k) - 1) * k
    n = k + m_blocks
    p61 = args.field == "p61"
    if not args.block_bytes:
        args.block_bytes = 65536 if p61 else 4096
    S = args.block_bytes // 4
    if p61:
        data = random_stripe_p61(k * (args.block_bytes // 8), device, seed=0x1234 + rank)
    else:
        data = random_stripe(k * S * args.batch, device, seed=0x1234 + rank)
    parity = torch.empty(data.numel() // k * m_blocks, dtype=data.dtype, device=device)
    if args.batch > 1:
        step = lambda: enc.encode_batch(data, parity, args.batch, stream=stream)  # noqa: E731
    else:
        step = lambda: enc.encode(data, parity, stream=stream)  # noqa: E731
    field = fastecc_amd.FIELD_GF_P61_SQUARED if p61 else fastecc_amd.FIELD_GF_FFF00001
    enc = fastecc_amd.Encoder(n, k, args.block_bytes, device=local, field=field)

    def tune(e):
        if args.plan:
            e.set_plan(args.plan)
        if args.slabs:
            e.set_option("slabs", args.slabs)
        for kv in args.option:
            name, value = kv.split("=")
            e.set_option(name, int(value))

    tune(enc)
    stream = torch.cuda.current_stream().cuda_stream

    # ---- N > 1: this rank's own measurement first, with no collective anywhere near it, THEN the process group under a timer.  RCCL has never run
    # under this file before the driver's first multi-GPU box: if the process group (or its first barrier) does not come up, rank 0 still prints
    # a line — its own timing of its stripe, times N — and says what it is. ----
    if dist_on(world):
        import torch.distributed as dist
        for _ in range(args.warmup):
            step()
        local_ms = time_steps(step, args.steps, torch.cuda.synchronize) / args.steps * 1e3

        def no_process_group():
            # Only what was measured: rank 0's own stripe on ONE GPU (no barrier, no max over ranks, the other ranks may never have run).
            # n_gpus says 1, the N-rank extrapolation sits in a key of its own, `complete` is false and the exit code is EXIT_NO_GROUP,
            # so that a failed RCCL start cannot be recorded as a measured N-GPU result.
            if rank == 0:
                bytes_per_encode = float(k + m_blocks) * args.block_bytes * args.batch
                one_gpu = bytes_per_encode / (local_ms * 1e-3) / 1e9
                print(json.dumps({
                    "metric": "encode GB/s (data+parity bytes / s); value = rank 0's OWN timing of its stripe on ONE GPU: the process group of %d ranks did "
                              "not come up within %d s, so no barrier, no max over ranks and no one-stripe mode could run" % (world, args.startup_timeout),
                    "value": round(one_gpu, 2), "value_kind": "rank0_only", "complete": False, "unit": "GB/s", "n_gpus": 1, "requested_gpus": world,
                    "steps": args.steps, "warmup": args.warmup,
                    "ms_per_step": round(local_ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64" if p61 else "u32",
                    "data": "synthetic", "config": {"workload": "one stripe on one GPU, k=2^%d x %d B blocks" % (args.log2k, args.block_bytes), "plan": enc.plan(),
                                                    "parallelism": "rank 0 alone (of %d requested ranks)" % world},
                    "collectives": "UNAVAILABLE (torch.distributed %s did not initialise or pass its first barrier within %d s)" % (backend, args.startup_timeout),
                    "replicas_extrapolated_GBps": round(world * one_gpu, 2),
                    "replicas_extrapolated_note": "rank 0's figure x %d ranks: NOT measured" % world,
                    "rank0_local_ms_per_step": round(local_ms, 4)}), flush=True)
            else:
                time.sleep(3)
            os._exit(EXIT_NO_GROUP)

        startup = threading.Timer(args.startup_timeout, no_process_group)
        startup.daemon = True
        if args.startup_timeout > 0:
            startup.start()
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("FASTECC_BENCH_TEST_STALL", "") == "startup" and rank == world - 1:
            time.sleep(1e6)  # test hook: one rank never joins the process group
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
        dist.barrier()
        torch.cuda.synchronize()
        startup.cancel()

    def barrier():
        if dist_on(world):
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if not dist_on(world):
            return x
        t = torch.tensor([x], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ---- the contract's timed region: W warm-up steps, then exactly K steps, nothing but the encode calls inside ----
    for _ in range(args.warmup):
        step()
    elapsed = max_over_ranks(time_steps(step, args.steps, barrier))
    # ---- a second, instrumented pass of K steps: HIP events around every kernel, on the stream the kernels run on ----
    enc.profile(True)
    enc.profile_reset()
    elapsed_prof = max_over_ranks(time_steps(step, args.steps, barrier))
    kernels = enc.profile_read()
    enc.profile(False)

    # ---- everything the line needs from the replica measurement, before anything that could stall ----
    roof = check = None
    if rank == 0:
        bytes_per_encode = float(k + m_blocks) * args.block_bytes * args.batch  # data + parity, RS.cpp:38
        ms_per_step = elapsed / args.steps * 1e3
        ms_per_step_prof = elapsed_prof / args.steps * 1e3
        value = world * bytes_per_encode / (ms_per_step * 1e-3) / 1e9
        # dominant kernel by total time; a launch reads its part of the stripe once and writes it once (the
        # library reports those algorithmic bytes per launch: the whole stripe, or one column slab of it)
        dom = max(kernels.items(), key=lambda kv: kv[1][0]) if kernels else None
        if dom:
            name, (ms_total, launches, nbytes) = dom
            avg_ms = ms_total / launches
            per_launch = nbytes / launches
            achieved = per_launch / (avg_ms * 1e-3) / 1e9
            kernel_ms_per_step = sum(v[0] for v in kernels.values()) / args.steps  # summed durations (kernels may overlap)
            per_block = args.block_bytes // 16 if p61 else S  # field elements per block
            log2m = args.log2k if args.log2m is None or m_blocks > k else args.log2m
            bfly = (args.log2k * (k / 2) + (log2m + 1) * (m_blocks / 2)) * per_block * args.batch / (ms_per_step * 1e-3) / 1e9
            headline = args.log2k == 19 and args.block_bytes == 4096 and not p61 and m_blocks == k and args.batch == 1 and not args.plan and not args.option
            # the committed counter files are quoted only while they describe this binary (tools/stamp_counters.py); otherwise the fields say STALE
            state = counters_state(loaded_kernels=list(kernels)) if headline else {"status": "not_applicable", "why": "counter files exist for the headline configuration only"}
            quoted = state["status"] == "ok"
            stale_note = None if quoted or not headline else "%s: %s" % (state["status"], state.get("why"))
            traffic = pmc_traffic(name) if quoted or not headline else None
            prof_ms = rocprof_avg_ms(name) if headline and quoted else None
            # `bound` comes from counters, not from the kernel's name: profiles/r05/pmc_valu_default_plan.json (tools/pmc_valu.py) holds, for the
            # default plan's three kernels at the headline size, VALU instructions issued per SIMD and cycle against the isolated butterfly
            # loop's rate (valu_issue_frac) and the algorithmic HBM rate against what a copy reaches (hbm_frac_achievable); the larger one names
            # the bound.  achieved / peak / frac stay the HBM figures the contract asks for.  Other sizes / plans / fields have no counter file:
            # their `bound` is labelled as assumed.
            cfg5 = p61 and args.log2k == 19 and args.block_bytes == 65536 and m_blocks == k and args.batch == 1 and not args.plan and not args.option
            pmc_file = (PMC_VALU if quoted else None) if headline else PMC_VALU_P61 if cfg5 else None
            pmc = pmc_valu(pmc_file) if pmc_file else None
            ev = (pmc or {}).get("kernels", {}).get(name)
            roof = {"bound": ev["bound"] if ev else ("valu" if "_mid" in name else "hbm"),
                    "bound_source": (pmc_file + " (rocprofv3 --pmc SQ_INSTS_VALU / GRBM_GUI_ACTIVE of this command and of the isolated butterfly loop, committed; "
                                                "not measured in this run)") if ev else ("assumed from the kernel's kind (%s)" % (stale_note or "no counter file for this size / plan / field")),
                    "committed_counters": state,
                    "bound_evidence": None if not ev else {k2: ev.get(k2) for k2 in ("valu_issue_frac", "cycles_per_valu_instruction", "hbm_frac_achievable", "clock_GHz",
                                                                                  "valu_busy_gfx94x_formula", "wave_cycles_split")},
                    "per_kernel_bound": None if not pmc else {kn: {k2: e[k2] for k2 in ("bound", "valu_issue_frac", "hbm_frac_achievable", "clock_GHz")}
                                                              for kn, e in pmc["kernels"].items()},
                    "kernel": name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                    "traffic_source": stale_note if traffic is None else "profiles/%s/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE of an "
                                                                   "earlier run of this command, corrected per MI355X_MICROARCH.md; not measured in this run)" % PROFILE_ROUND,
                    "avg_kernel_ms": round(avg_ms, 4), "alg_bytes_per_launch": per_launch,
                    "frac_rocprof": None if not prof_ms else round(per_launch / (prof_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                    "rocprof_avg_kernel_ms": None if not prof_ms else round(prof_ms, 4),
                    "rocprof_source": stale_note if not prof_ms else ROCPROF_STATS + " (rocprofv3 --kernel-trace --stats of this command, committed; not measured in this run)",
                    "encode": {"ms_per_step": round(ms_per_step, 4), "sum_of_kernel_ms_per_step": round(kernel_ms_per_step, 4),
                               "achieved": round(bytes_per_encode / (ms_per_step * 1e-3) / 1e9, 1),
                               "frac": round(bytes_per_encode / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                               "hbm_trips": int(round(sum(v[1] for v in kernels.values()) / max(args.steps, 1)))},
                    "valu": {"what": "radix-2 butterflies per second over the whole encode (2*log2(k)*k/2 per element column, "
                                     "plus k/2 butterfly-equivalents for the per-block factor multiply)",
                             "achieved_Gbfly_per_s": round(bfly, 1), "microbench_peak_Gbfly_per_s": None if p61 else VALU_PEAK_GBFLY,
                             "frac": None if p61 else round(bfly / VALU_PEAK_GBFLY, 4),
                             # the isolated loop runs at 2.39 GHz (no HBM traffic, 1055 W); the encode at ~2.0 GHz (1358 of 1400 W): the same yardstick at
                             # the clock the encode's kernels ran at, and the instruction-level version of it (all VALU instructions of the three kernels
                             # at the isolated loop's issue rate)
                             "microbench_clock_GHz": None if p61 else VALU_PEAK_CLOCK_GHZ,
                             "encode_clock_GHz": None if not pmc else round(pmc["encode"]["cycles"] / pmc["encode"]["sum_of_kernel_ms"] / 1e6, 3),
                             "frac_at_encode_clock": None if not pmc or p61 else round(bfly / (VALU_PEAK_GBFLY * pmc["encode"]["cycles"] / pmc["encode"]["sum_of_kernel_ms"] / 1e6
                                                                                        / VALU_PEAK_CLOCK_GHZ), 4),
                             "issue_floor_frac": None if not pmc else pmc["encode"]["valu_floor_frac"]},
                    "per_kernel_avg_ms": {kn: round(v[0] / v[1], 4) for kn, v in sorted(kernels.items())},
                    "launches_per_step": {kn: v[1] // args.steps for kn, v in sorted(kernels.items())}}
        if not args.no_parity_check and args.batch == 1 and m_blocks == k:
            try:
                check = parity_check_p61(data, parity, k, args.block_bytes) if p61 else parity_check(enc, args.log2k, args.block_bytes, device)
            except Exception as e:  # noqa: BLE001
                check = {"status": "error", "why": repr(e)}

    # ---- who measured: one record per rank (the driver's scaling line must be able to show N distinct GPUs) ----
    def identity():
        rec = {"rank": rank, "local_rank": local, "device_index": torch.cuda.current_device()}
        try:
            pr = torch.cuda.get_device_properties(device)
            rec["name"] = pr.name
            rec["uuid"] = str(getattr(pr, "uuid", ""))
            rec["pci_bus_id"] = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0))
            rec["compute_units"] = pr.multi_processor_count
            rec["hbm_GiB"] = round(pr.total_memory / 2**30, 1)
            rec["gcn_arch"] = getattr(pr, "gcnArchName", "")
            rec["can_access_peer"] = [bool(j == local or torch.cuda.can_device_access_peer(local, j)) for j in range(torch.cuda.device_count())]
        except Exception as e:  # noqa: BLE001
            rec["error"] = repr(e)
        for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
            if os.environ.get(var) is not None:
                rec[var] = os.environ[var]
        return rec

    devices = [identity()]
    dist_info = {"world_size": 1, "backend": None}
    if dist_on(world):
        try:
            every_rec = [None] * world
            dist.all_gather_object(every_rec, devices[0])
            devices = every_rec
            dist_info = {"world_size": dist.get_world_size(), "backend": str(dist.get_backend()),
                         "rccl": ".".join(str(v) for v in torch.cuda.nccl.version()) if backend == "nccl" else None,
                         "distinct_gpus": len({(d or {}).get("uuid") or (d or {}).get("pci_bus_id") for d in devices})}
        except Exception as e:  # noqa: BLE001
            dist_info = {"world_size": world, "backend": backend, "error": repr(e)}

    emitted = threading.Event()
    other_paths_result = {}

    def emit(one, cabi=None, cpu=None):
        if rank != 0 or emitted.is_set():
            return
        emitted.set()
        code = ("(n,k)=(2^%d,2^%d), %d-byte blocks" % (args.log2k + 1, args.log2k, args.block_bytes)) if m_blocks == k else \
               ("k=2^%d data + %d parity blocks, %d-byte blocks" % (args.log2k, m_blocks, args.block_bytes))
        fieldname = "GF((2^61-1)^2)" if p61 else "GF(0xFFF00001)"
        replicas = {"value": round(value, 2), "unit": "GB/s", "ms_per_step": round(ms_per_step, 4), "scaling": "weak",
                    "what": "%d independent stripe(s), one per GPU, no collective in the timed region (the mode for many stripes: "
                            "different stripes are independent jobs)" % world}
        # N > 1: `value` is BASELINE configs[3] — ONE stripe over all GPUs, the parity handed over block-distributed (all-to-all); the
        # replica figure stays in the line as `replicas`.  If that mode did not complete, the line says so and falls back to the replicas.
        a2a = (one or {}).get("all_to_all") if world > 1 else None
        kind = "one_stripe_per_gpu"
        if a2a and "ms_per_stripe" in a2a:
            kind = "one_stripe_all_to_all"
            v, ms, scaling = a2a["GBps"], a2a["ms_per_stripe"], "strong"
            which = ("ONE stripe over %d GPUs: column-slab encode + RCCL all-to-all into block-distributed parity (one_stripe.all_to_all)" % world)
            workload = ("RS encode k=2^%d data -> %d parity blocks, %d B blocks, %s, ONE %.0f MiB stripe sharded over %d GPUs in column slabs "
                        "(HBM-resident), parity block-distributed by an all-to-all over xGMI" % (args.log2k, m_blocks, args.block_bytes, fieldname,
                                                                                               k * args.block_bytes / 2**20, world))
            par = "one stripe, %d column slabs, all-to-all of the parity (strong scaling)" % world
        else:
            v, ms, scaling = value, ms_per_step, "weak"
            if world > 1:
                kind = "replicas_fallback"  # a different quantity than the N > 1 headline (weak, not strong scaling): `complete` is false
            which = "one stripe per GPU" if world == 1 else "REPLICAS (one independent stripe per GPU): the one-stripe all-to-all mode did not complete, see one_stripe"
            workload = ("RS encode k=2^%d data -> %d parity blocks, %d B blocks, %s, one %.0f MiB stripe per GPU, HBM-resident, out of place"
                        % (args.log2k, m_blocks, args.block_bytes, fieldname, k * args.block_bytes / 2**20))
            par = "%d independent stripe(s), one per GPU, no collective" % world
        line = {
            "metric": "encode GB/s at %s (data+parity bytes / s); value = %s" % (code, which),
            "value": round(v, 2), "value_kind": kind,
            "complete": kind != "replicas_fallback" and (one or {}).get("complete", True) is not False,
            "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 4), "ms_per_step_instrumented": round(ms_per_step_prof, 4),
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "u64" if p61 else "u32", "data": "synthetic",
            "config": {"workload": workload, "stripes_per_step": args.batch, "plan": enc.plan(), "parallelism": par},
            "data_only_GBps": round(v / 2, 2),
            # the same ONE stripe without any exchange (parity left in column slabs): read a 1 -> N curve with `value` (exchange included at
            # N > 1, none at N = 1) and with this (no exchange at any N)
            "compute_only_GBps": ((one or {}).get("compute_only") or {}).get("GBps"),
            "compute_only_ms_per_stripe": ((one or {}).get("compute_only") or {}).get("ms_per_stripe"),
            "exchange_only_ms_per_stripe": ((one or {}).get("exchange_only") or {}).get("ms_per_stripe") if world > 1 else None,
            # what the links gave in THIS run (exchange_only) beside what DESIGN.md §8 assumes for one xGMI link per direction
            "link_peak_GBps": (one or {}).get("link_peak_GBps") if world > 1 else None,
            "link_peak_source": (one or {}).get("link_peak_source") if world > 1 else None,
            "link_peak_assumed_GBps": (one or {}).get("link_peak_assumed_GBps") if world > 1 else None,
            "expected_shape": None if world == 1 else (
                "N=2 is SLOWER than N=1 by construction: one xGMI link carries a quarter of the stripe's parity (DESIGN.md §8); compare compute_only_GBps"
                if world == 2 else "value includes the all-to-all of the parity over xGMI; compute_only_GBps is the same stripe without it (DESIGN.md §8)"),
            "replicas": replicas,
            "parity_check": check,
            "roofline": roof, "cpu_baseline": cpu,
            "one_stripe": one,
            "devices": devices, "distributed": dist_info,
        }
        if _ACTIVE_TEST_HOOKS:
            line["test_hooks_active"] = {v: os.environ[v] for v in _ACTIVE_TEST_HOOKS}
        if other_paths_result:
            line["other_paths"] = other_paths_result
        if p61:
            line["parity_pin"] = ("no upstream code exists for this field: the HIP path is pinned to this repository's own oracle and "
                                  "Python big-integer goldens (tests/test_gpu_p61.py), not to the reference")
        if cabi is not None:
            line["one_stripe_c_abi"] = cabi
        print(json.dumps(line), flush=True)

    # N > 1: the one-stripe modes below are the first thing in this file that needs every rank to make progress together inside
    # RCCL transfers of whole parity slabs.  Whatever happens there, the measurements already taken must reach the driver:
    #   * every mode runs under a per-mode timer (--mode-timeout); the whole section under --sharded-timeout;
    #   * at expiry rank 0 prints the line with `one_partial` as it stands (the replica measurement, every finished mode — `value` is
    #     all_to_all, which runs first among the exchanging modes) and every rank leaves.
    one_partial = {}
    armed = {"timer": None}

    def give_up(where, seconds):
        one_partial.setdefault(where, {})
        if isinstance(one_partial[where], dict) and "ms_per_stripe" not in one_partial[where]:
            one_partial[where] = {"error": "no progress within %d s (stalled); the line was printed by the watchdog" % seconds}
        one_partial["stalled_in"] = where
        one_partial["complete"] = False
        emit(one_partial)
        sys.stdout.flush()
        if rank != 0:
            time.sleep(3)  # let rank 0 print before its collectives see a peer disappear
        os._exit(EXIT_WATCHDOG)

    class watch:  # noqa: N801 — `with watch(name):` around one mode
        def __init__(self, name):
            self.name = name

        def __enter__(self):
            if dist_on(world) and args.mode_timeout > 0:
                armed["timer"] = threading.Timer(args.mode_timeout, give_up, args=(self.name, args.mode_timeout))
                armed["timer"].daemon = True
                armed["timer"].start()

        def __exit__(self, *exc):
            if armed["timer"] is not None:
                armed["timer"].cancel()
                armed["timer"] = None
            return False

    watchdog = None
    if dist_on(world) and args.sharded_timeout > 0:
        watchdog = threading.Timer(args.sharded_timeout, give_up, args=("one_stripe_section", args.sharded_timeout))
        watchdog.daemon = True
        watchdog.start()

    # ---- BASELINE configs[3]: ONE stripe over the ranks (strong scaling) — column-slab compute, three ways to hand over the parity ----
    one = None
    unit = 8 if p61 else 4  # bytes per tensor word
    words = args.block_bytes // unit
    shardable = (not args.no_sharded and args.batch == 1 and m_blocks == k and words % world == 0 and (not p61 or world > 1)
                 and (args.block_bytes // world) % (16 if p61 else 4) == 0 and k % world == 0)
    if shardable:
        try:
            one = one_stripe_modes(args, fastecc_amd, field, device, local, rank, world, backend, stream, tune, barrier, max_over_ranks, k, n, p61,
                                   one_partial, watch)
        except Exception as e:  # noqa: BLE001 — the replica number above and the modes already measured must survive a failure here
            one_partial["error"] = repr(e)
            one = one_partial

    if watchdog is not None:
        watchdog.cancel()

    cabi = cpu = None
    if rank == 0:
        if (world == 1 and torch.cuda.device_count() > 1 and not args.no_sharded and not p61 and args.batch == 1 and m_blocks == k
                and not os.environ.get("FASTECC_BENCH_NO_CABI_SHARDED")):
            # the single-process form of configs[3] on every visible GPU, in a child (its own HIP contexts; a failure
            # or a hang costs this entry, not the line)
            import subprocess
            cmd = [sys.executable, os.path.abspath(__file__), "--cabi-sharded-child", "--steps", str(min(args.steps, 10)),
                   "--log2k", str(args.log2k), "--block-bytes", str(args.block_bytes)]
            def last_json(text):
                if isinstance(text, bytes):
                    text = text.decode("utf-8", "replace")
                for ln in reversed((text or "").splitlines()):
                    if ln.startswith("{"):
                        try:
                            return json.loads(ln)
                        except ValueError:  # a line cut off by the fault that ended the child
                            continue
                return None

            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=180)
                cabi = last_json(r.stdout) or {"error": (r.stderr or r.stdout)[-400:]}
                if not cabi.get("complete") and "error" not in cabi:
                    cabi["incomplete"] = "child exited with code %d after these measurements: %s" % (r.returncode, (r.stderr or "")[-300:])
            except subprocess.TimeoutExpired as e:
                cabi = last_json(e.stdout) or {}
                cabi["incomplete"] = "child timed out after these measurements"
            except Exception as e:  # noqa: BLE001
                cabi = {"error": repr(e)}
        if not args.no_cpu_baseline and world == 1:
            try:
                cpu = (cpu_baseline_p61 if p61 else cpu_baseline)(args.cpu_log2k or args.log2k, args.block_bytes)
            except Exception as e:  # the baseline is a report, never a reason to lose the GPU number
                cpu = {"error": repr(e)}
    if rank == 0 and world == 1 and not args.no_other_paths and not p61 and args.batch == 1 and m_blocks == k and not args.plan and not args.option:
        # in a child process: a fault in one of these paths must not cost the line its headline number
        import subprocess
        cmd = [sys.executable, os.path.abspath(__file__), "--other-paths-child", "--log2k", str(args.log2k), "--block-bytes", str(args.block_bytes)]
        # (not this process's OpenMP settings: spinning worker threads — wanted for the CPU baseline above — eat the container's CPU quota, and the
        #  HIP runtime's own threads then starve: the child's host-link copies ran at 40 instead of 57 GB/s and its pipeline timing jumped 80 <-> 100 ms)
        child_env = dict(os.environ, OMP_WAIT_POLICY="passive", OMP_NUM_THREADS=str(min(8, usable_cpus())))
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=child_env)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            other_paths_result.update(json.loads(lines[-1]) if lines else {"error": (r.stderr or r.stdout)[-400:]})
        except Exception as e:  # noqa: BLE001
            other_paths_result.update({"error": repr(e)})
    emit(one, cabi, cpu)
    enc.close()
    if dist_on(world):
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
