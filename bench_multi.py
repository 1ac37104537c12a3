"""bench_multi.py - the N > 1 half of bench.py (BASELINE configs[3] / [4]): the one-stripe modes over the ranks, the single-process C-ABI form in a
child, and the test hooks that let a 1-GPU box run this control flow (gloo, one-rank RCCL groups, injected faults).  Imported by bench.py; the
driver's command line stays `python bench.py --gpus N ...`."""
import json
import os
import sys
import time

import torch

from bench_common import ROOT, random_stripe, random_stripe_p61, splitmix_window, time_steps  # noqa: F401

# Test hook (tests/test_gpu_bench_rccl_one_rank.py): a one-rank process group is brought up and used as if the job had several ranks, so that
# the RCCL-specific lines of this file (device-side reductions, communicators per mode, object gathers) run against the real library on a
# 1-GPU box.  Together with FASTECC_SHARDING_FORCE_COLLECTIVES the one_stripe modes then issue their collectives too.  Never set by the driver.
ONE_RANK_GROUP = os.environ.get("FASTECC_BENCH_TEST_ONE_RANK_GROUP", "") == "1"
# Exit codes of an N > 1 run that printed a line but did not finish: the line carries "complete": false, and the launcher sees a failure.
EXIT_NO_GROUP = 3   # the process group never came up: the line holds rank 0's own one-GPU timing (n_gpus = 1)
EXIT_WATCHDOG = 4   # a one-stripe mode stalled: its timer printed the line with everything measured so far

ACTIVE_TEST_HOOKS = [v for v in ("FASTECC_BENCH_TEST_ONE_RANK_GROUP", "FASTECC_BENCH_TEST_STALL", "FASTECC_BENCH_BACKEND", "FASTECC_SHARDING_FORCE_COLLECTIVES")
                      if os.environ.get(v)]
if ACTIVE_TEST_HOOKS:  # a stray exported variable must not change what the driver measures silently: say so, loudly, on every rank
    print("[bench.py] TEST HOOKS ACTIVE in the environment: %s - this is NOT a production measurement" % ", ".join(
        "%s=%s" % (v, os.environ[v]) for v in ACTIVE_TEST_HOOKS), file=sys.stderr, flush=True)


def dist_on(world):
    return world > 1 or ONE_RANK_GROUP


def cabi_sharded_child(args):
    """Single process, all visible GPUs: BASELINE configs[3] through fastecc_create_sharded (csrc/sharded.hip).
    Prints one JSON object; run by the parent bench in a child process so that a failure cannot take the headline
    number with it."""
    import fastecc_amd
    G = torch.cuda.device_count()
    k, bb = 1 << args.log2k, args.block_bytes or 4096
    S = bb // 4
    while G > 1 and S % (G * 32):
        G -= 1
    ids = list(range(G))
    w = S // G
    out = {"n_gpus": G, "slab_bytes_per_block": 4 * w, "steps": args.steps}
    slabs, pslabs = [], []
    for g in ids:
        dev = torch.device("cuda", g)
        slabs.append(random_stripe(k * w, dev, seed=0x1234 + g))
        pslabs.append(torch.empty(k * w, dtype=torch.int32, device=dev))
    torch.cuda.set_device(0)
    parity = torch.empty(k * S, dtype=torch.int32, device="cuda:0")
    hx = torch.empty(k * S, dtype=torch.int32).pin_memory()
    hp = torch.empty(k * S, dtype=torch.int32).pin_memory()
    hx.copy_(torch.cat([t.view(k, w).cpu() for t in slabs], dim=1).reshape(-1))
    enc = fastecc_amd.ShardedEncoder(2 * k, k, bb, ids)
    stream = torch.cuda.current_stream().cuda_stream

    def sync_all():
        for g in ids:
            torch.cuda.synchronize(g)

    def timed(fn):
        for _ in range(2):
            fn()
        sync_all()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fn()
        sync_all()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        return {"ms_per_stripe": round(ms, 4), "GBps": round(2.0 * k * bb / (ms * 1e-3) / 1e9, 2)}

    def progress():
        # one line per finished measurement: the parent keeps the last complete one, so a device fault or a hang in a
        # later mode (peer copies have never run on this code before the first multi-GPU box) costs only that mode
        print(json.dumps(out), flush=True)

    def gathered_ok():
        return all(torch.equal(parity.view(k, S)[:, g * w:(g + 1) * w].cpu(), pslabs[g].view(k, w).cpu()) for g in ids)

    out["plan"] = enc.plan()
    out["compute_only"] = timed(lambda: enc.encode_sharded(slabs, pslabs, None, stream=stream))
    progress()
    for mode, name in ((2, "kernel"), (1, "copy_engine")):
        enc.set_option("gather_mode", mode)
        for sub in (1, 2, 4):
            enc.set_option("sub_slabs", sub)
            try:
                parity.zero_()
                out["with_gather_%s_sub%d" % (name, sub)] = timed(lambda: enc.encode_sharded(slabs, pslabs, parity, stream=stream))
                # correctness of what was just timed: the gathered parity equals the slabs that stayed on their GPUs
                out["with_gather_%s_sub%d" % (name, sub)]["gather_check"] = "ok" if gathered_ok() else "FAILED"
            except Exception as e:  # noqa: BLE001
                out["with_gather_%s_sub%d" % (name, sub)] = {"error": repr(e)}
            progress()
    checks = [v.get("gather_check") for kname, v in out.items() if kname.startswith("with_gather_") and "gather_check" in v]
    out["gather_check"] = "ok" if checks and all(c == "ok" for c in checks) else "FAILED"
    # the block-distributed form (fastecc_encode_sharded_blocks): GPU g ends with parity blocks [g*k/G, (g+1)*k/G) whole — an all-to-all over
    # the peers instead of a gather into the root — from data in slabs, and from block-distributed data (mirror transpose in front)
    if k % G == 0:
        rows = k // G
        pblocks = [torch.empty(rows * S, dtype=torch.int32, device="cuda:%d" % g) for g in ids]
        dblocks = [hx.view(k, S)[g * rows:(g + 1) * rows].to("cuda:%d" % g).reshape(-1) for g in ids]

        def blocks_ok():
            return all(torch.equal(pblocks[j].view(rows, S)[:, g * w:(g + 1) * w].cpu(), pslabs[g].view(k, w)[j * rows:(j + 1) * rows].cpu())
                       for g in ids for j in ids)

        enc.encode_sharded(slabs, pslabs, None, stream=stream)
        sync_all()
        for mode, name in ((2, "kernel"), (1, "copy_engine")):
            enc.set_option("gather_mode", mode)
            for sub in (1, 2, 4):
                enc.set_option("sub_slabs", sub)
                for as_blocks, tag in ((False, "all_to_all"), (True, "all_to_all_in_out")):
                    key = "%s_%s_sub%d" % (tag, name, sub)
                    try:
                        for t in pblocks:
                            t.zero_()
                        out[key] = timed(lambda: enc.encode_sharded_blocks(dblocks if as_blocks else slabs, pblocks, data_is_blocks=as_blocks, stream=stream))
                        out[key]["check"] = "ok" if blocks_ok() else "FAILED"
                    except Exception as e:  # noqa: BLE001
                        out[key] = {"error": repr(e)}
                    progress()
        del pblocks, dblocks
    enc.set_option("gather_mode", 1)
    enc.set_option("sub_slabs", 2)
    try:
        out["root_resident_stripe"] = timed(lambda: enc.encode(parity, parity, stream=stream))
        out["host_pinned_stripe"] = timed(lambda: enc.encode(hx, hp, mem=fastecc_amd.MEM_HOST_PINNED, stream=stream))
        out["host_pinned_stripe"]["what"] = "pinned host stripe in, parity out: every GPU moves its slab over its own host link"
    except Exception as e:  # noqa: BLE001
        out["stripe_modes_error"] = repr(e)
    out["complete"] = True
    progress()
    enc.close()


LINK_GBPS_PER_DIRECTION_ASSUMED = 76.8  # one xGMI link: ~153.6 GB/s both directions together ("7 links x ~153 GB/s"); an ASSUMPTION —
# Order of the one-stripe modes = order of importance: `value` comes from all_to_all, so it runs right after the exchange-free baseline; the
# gather-to-root (the mode DESIGN.md section 8 itself says cannot scale) runs last, where a failure or a stall in it can cost nothing else.
ONE_STRIPE_MODES = ("compute_only", "all_to_all", "exchange_only", "all_to_all_in_out", "gather_to_root")
OWN_GROUP_MODES = ("all_to_all_in_out", "gather_to_root")  # modes after `value`: each on its own communicator (dist.new_group)


def injected_fault(mode, rank, world):
    """Test hook FASTECC_BENCH_TEST_STALL=<mode>[:stall|:raise|:raise_all] — the LAST rank stalls (never enters the mode's collectives) or raises
    inside `mode` (raise_all: every rank raises); "1" / "all" = the last rank never reaches the one-stripe modes at all.  tests/test_gpu_sharded.py drives it with gloo on one GPU."""
    spec = os.environ.get("FASTECC_BENCH_TEST_STALL", "")
    name, _, how = spec.partition(":")
    if not spec or (rank != world - 1 and how != "raise_all"):
        return
    if name in ("1", "all"):
        name = "before_the_modes"
    if name != mode:
        return
    if how in ("raise", "raise_all"):
        raise RuntimeError("FASTECC_BENCH_TEST_STALL: injected failure in %s on rank %d" % (mode, rank))
    time.sleep(1e6)


def one_stripe_modes(args, fastecc_amd, field, device, local, rank, world, backend, stream, tune, barrier, max_over_ranks, k, n, p61, out, watch):
    """BASELINE configs[3]: ONE (n,k) stripe over the ranks.  The compute is always the column-slab encode (rank r holds words
    [r*S/G, (r+1)*S/G) of every block: no exchange inside the transform); the modes differ in what happens to the parity:

        compute_only        it stays in slabs (no exchange at all)
        all_to_all          block-distributed: rank g ends with parity blocks [g*M/G, (g+1)*M/G) whole (RCCL all-to-all: 1/G^2 of the
                            stripe per link and direction, no hot spot) — the line's `value` at N > 1
        exchange_only       all_to_all's exchange without the encode (what the links alone allow: the measured link roofline)
        all_to_all_in_out   the data arrives block-distributed as well (whole data blocks per rank): the mirror transpose in front
        gather_to_root      full blocks on rank 0 (RCCL gather; bound by the root's links: (G-1)/G of the stripe enters one GPU)

    Every mode: W warm-up calls, then exactly K calls between barriers, max over ranks.  A mode that raises costs only itself: the error is
    recorded and the next mode runs (the two modes after `value` on communicators of their own, so a broken one is not reused).  A mode that
    STALLS cannot be cancelled from Python (a rank blocked inside a collective), so `watch(name)` arms a per-mode timer whose expiry prints the
    line with everything measured so far and ends the job: by the order above a stall can only cost modes less important than its own.
    `out` is filled as the modes finish (the timer reads it).  The stripe is the splitmix64(0x1234) one, so what was timed is checked
    against the unmodified reference's parity hash (main.cpp:202-212) where a golden value exists."""
    import torch.distributed as dist
    from fastecc_amd import sharding
    unit = 8 if p61 else 4
    words = args.block_bytes // unit
    S32 = args.block_bytes // 4
    w = words // world
    sub = sharding.sub_slab_count(w, args.sub_slabs, unit // 4)
    wsub = w // sub
    kg = k // world
    gloo = backend != "nccl"
    stripe_bytes = float(k) * args.block_bytes  # data = parity bytes of the (2k,k) stripe
    G = world
    out.update({"what": "ONE stripe of k=2^%d x %d B blocks in %d column slabs of %d B per block, one per rank, each resident as %d contiguous sub-slab(s) "
                        "(the exchange of one sub-slab runs on side streams under the next sub-slab's encode).  compute_only: the parity stays in slabs; "
                        "all_to_all: block-distributed result, rank g holds parity blocks [g*M/G, (g+1)*M/G) whole; exchange_only: all_to_all without the "
                        "encode; all_to_all_in_out: the data block-distributed as well (mirror transpose in front); gather_to_root: RCCL gather into full "
                        "blocks on rank 0 (no pack, the root encodes straight into the full blocks)"
                        % (args.log2k, args.block_bytes, world, args.block_bytes // world, sub),
                "scaling": "strong", "sub_slabs": sub, "mode_order": list(ONE_STRIPE_MODES),
                "link_model": "link_bytes_in_max_rank = bytes entering the busiest rank per stripe; link_GBps_in_max_rank = that / ms_per_stripe; "
                              "link_peak_GBps = the rate exchange_only reached in THIS run (the all-to-all of the same bytes with no encode around it: "
                              "what RCCL over these links delivers), link_roofline_frac = rate / that; link_peak_assumed_GBps = min(G-1,7) links x %.1f GB/s "
                              "per direction (half of ~153.6 GB/s per link, never measured here)" % LINK_GBPS_PER_DIRECTION_ASSUMED})
    with watch("setup"):
        injected_fault("before_the_modes", rank, world)
        senc = fastecc_amd.Encoder(n, k, args.block_bytes // world // sub, device=local, field=field)
        tune(senc)
        out["plan"] = senc.plan()
        # this rank's slab, resident as `sub` contiguous column sub-slabs [sub][k][wsub] (what a scatter delivers), and its whole data blocks
        if p61:
            slab = random_stripe_p61(k * w, device, seed=0x5EED + rank).view(sub, k, wsub)
            blocks = None  # (the block-distributed input needs every rank to derive the same stripe: GF(0xFFF00001) only)
        else:
            slab = torch.stack([splitmix_window(device, S32, 0, k, rank * w + h * wsub, wsub) for h in range(sub)])
            blocks = splitmix_window(device, S32, rank * kg, kg, 0, S32)
        pslab = torch.empty_like(slab)
    wsp_g, wsp_a, wsp_b, wsp_x = {}, {}, {}, {}
    # gather_to_root: the root keeps its slab at the full block pitch ([k][world*w] arrays, its own columns filled): with "row_pitch_words"
    # its encoder reads the sub-slab there and writes the parity straight into the full parity blocks, so the root's part is neither sent
    # nor re-interleaved (GF(0xFFF00001) contexts; the 64-bit field keeps the contiguous form).
    in_place = rank == 0 and not p61
    penc = None
    gslab = slab
    if in_place:
        penc = fastecc_amd.Encoder(n, k, args.block_bytes // world // sub, device=local, field=field)
        tune(penc)
        penc.set_option("row_pitch_words", words)
        full_data = torch.zeros((k, world, sub, wsub), dtype=slab.dtype, device=device)
        full_data[:, 0] = slab.permute(1, 0, 2)
        gslab = full_data[:, 0].permute(1, 0, 2)  # [sub, k, wsub] views, row stride = the full block

    def enc_fn(d, o):
        (senc if d.is_contiguous() else penc).encode(d, o, stream=torch.cuda.current_stream().cuda_stream)

    def encode_all():
        for h in range(sub):
            senc.encode(slab[h], pslab[h], stream=stream)

    groups = {}  # mode -> its own process group (None = the default one)

    def make(name):
        grp = groups.get(name)
        if name == "compute_only":
            return encode_all
        if name == "all_to_all":
            return lambda: sharding.encode_all_to_all(slab, enc_fn, k, workspace=wsp_a, collective_on_host=gloo, group=grp)
        if name == "exchange_only":
            return lambda: sharding.encode_all_to_all(slab, lambda d, o: None, k, workspace=wsp_x, collective_on_host=gloo, group=grp)
        if name == "all_to_all_in_out":
            if blocks is None:
                return None
            return lambda: sharding.encode_all_to_all(blocks, enc_fn, k, data_is_blocks=True, sub_slabs=sub, workspace=wsp_b, collective_on_host=gloo,
                                                      group=grp)
        return lambda: sharding.encode_sub_slabs_and_gather(gslab, enc_fn, k, dst=0, workspace=wsp_g, collective_on_host=gloo, root_in_place=in_place,
                                                            group=grp)

    link_in = {"compute_only": 0.0, "gather_to_root": (G - 1) / G * stripe_bytes, "all_to_all": (G - 1) / G**2 * stripe_bytes,
               "all_to_all_in_out": 2 * (G - 1) / G**2 * stripe_bytes, "exchange_only": (G - 1) / G**2 * stripe_bytes}
    def everyone_ok(ok):
        """The ranks agree (default group) whether a mode's warm-up went through everywhere: a rank that raised must not go on to the next
        mode's collectives while the others enter this mode's barriers."""
        if not dist_on(world):
            return ok
        t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device="cpu" if gloo else device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item() == 1.0)

    for name in ONE_STRIPE_MODES:
        try:
            with watch(name):
                if dist_on(world) and name in OWN_GROUP_MODES:
                    # every rank takes part in every new_group call, in this order; a mode that cannot get a communicator of its own
                    # runs on the default one
                    try:
                        groups[name] = dist.new_group(backend=backend)
                    except Exception as e:  # noqa: BLE001
                        out.setdefault("notes", []).append("%s: dist.new_group failed (%r), running on the default group" % (name, e))
                fn = make(name)
                if fn is None:
                    continue
                err = None
                try:
                    injected_fault(name, rank, world)
                    for _ in range(max(1, args.warmup)):
                        fn()
                    torch.cuda.synchronize()
                except Exception as e:  # noqa: BLE001
                    err = e
                if not everyone_ok(err is None):
                    raise err if err is not None else RuntimeError("%s failed on another rank" % name)
                ms = max_over_ranks(time_steps(fn, args.steps, barrier)) / args.steps * 1e3
            out[name] = {"ms_per_stripe": round(ms, 4), "GBps": round(2.0 * stripe_bytes / (ms * 1e-3) / 1e9, 2), "link_bytes_in_max_rank": int(link_in[name])}
        except Exception as e:  # noqa: BLE001 — costs this mode only (ranks left inside its collective by a one-sided failure end at the timer)
            out[name] = {"error": repr(e)}
    # ---- the link figures: every exchanging mode against what the bare exchange reached in this run ----
    if world > 1:
        assumed = min(G - 1, 7) * LINK_GBPS_PER_DIRECTION_ASSUMED
        xo = out.get("exchange_only") or {}
        measured = link_in["exchange_only"] / (xo["ms_per_stripe"] * 1e-3) / 1e9 if "ms_per_stripe" in xo else None
        out["link_peak_GBps"] = None if measured is None else round(measured, 1)
        out["link_peak_source"] = "exchange_only of this run" if measured is not None else "not measured (exchange_only did not complete)"
        out["link_peak_assumed_GBps"] = round(assumed, 1)
        for name in ONE_STRIPE_MODES:
            rec = out.get(name) or {}
            if "ms_per_stripe" in rec and link_in[name]:
                rate = link_in[name] / (rec["ms_per_stripe"] * 1e-3) / 1e9
                rec["link_GBps_in_max_rank"] = round(rate, 1)
                rec["link_roofline_frac"] = None if measured is None else round(rate / measured, 4)
                rec["link_frac_of_assumed_peak"] = round(rate / assumed, 4)
    # ---- what was timed is also right (collectives on the default group; a failure here costs the checks, not the timings) ----
    checks = {}
    try:
        with watch("checks"):
            encode_all()
            torch.cuda.synchronize()
            mine_a = wsp_a.get("parity_sub")
            ok_local = mine_a is not None and bool(torch.equal(mine_a, pslab))
            full = wsp_g.get("parity_full")
            if rank == 0 and full is not None:  # the root's own columns of the gathered blocks = its slab
                ok_local = ok_local and bool(torch.equal(full[:, :w], pslab.permute(1, 0, 2).reshape(k, w)))
            flag = torch.tensor([1.0 if ok_local else 0.0], dtype=torch.float64, device="cpu" if gloo else device)
            if dist_on(world):
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            checks["slabs_equal_compute_only_on_every_rank"] = bool(flag.item() == 1.0)
            for key, wsp in (("all_to_all", wsp_a), ("all_to_all_in_out", wsp_b)):
                mineb = wsp.get("parity_blocks")
                if mineb is None:
                    continue
                if dist_on(world):
                    piece = mineb.cpu() if gloo else mineb
                    got = [torch.empty_like(piece) for _ in range(world)] if rank == 0 else None
                    dist.gather(piece, gather_list=got, dst=0)
                else:
                    got = [mineb]
                if rank == 0:
                    whole = torch.cat([g.to(device) for g in got], dim=0)  # [M, S]: every rank's whole parity blocks in block order
                    res = {"equals_gather_to_root": bool(torch.equal(whole, full)) if full is not None else None}
                    if not p61:
                        try:
                            from oracle import Oracle
                            with open(os.path.join(ROOT, "tests", "golden", "golden_hashes.json")) as f:
                                gold = json.load(f)
                            want = [c for c in gold.get("survey_appendix_b", []) + gold.get("cases", [])
                                    if c.get("input") == "splitmix" and c.get("log2N") == args.log2k and c.get("block_bytes") == args.block_bytes]
                            if want:
                                import numpy as np
                                h = Oracle().hash(whole.cpu().numpy().view(np.uint32))
                                res.update({"reference_parity_hash": h, "expected": want[0]["hash_parity"], "status": "ok" if h == want[0]["hash_parity"] else "FAILED"})
                        except Exception as e:  # noqa: BLE001
                            res["hash_error"] = repr(e)
                    else:
                        # the 64-bit field has no upstream output: the first element column of the block-distributed parity (rank 0's own
                        # data column) re-encoded by this repository's CPU oracle — the gate of configs[4]'s N > 1 line
                        try:
                            import numpy as np
                            from oracle import OracleP61
                            x = np.ascontiguousarray(slab[0][:, 0:2].cpu().numpy().view(np.uint64))
                            want = OracleP61().encode(x)
                            have = whole[:, 0:2].cpu().numpy().view(np.uint64)
                            res["oracle_column"] = "ok" if want.shape == have.shape and np.array_equal(want, have) else "FAILED"
                            res["oracle_column_what"] = "element column 0 of all %d parity blocks against oracle/fastecc_oracle_p61.c" % whole.shape[0]
                            if res["oracle_column"] != "ok":
                                res["status"] = "FAILED"
                        except Exception as e:  # noqa: BLE001
                            res["oracle_column"] = "error: %r" % e
                    if "status" not in res:
                        res["status"] = "unchecked (no golden hash at this size, no gathered copy)" if res["equals_gather_to_root"] is None else \
                                        "ok" if res["equals_gather_to_root"] else "FAILED"
                    checks[key] = res
                    del whole
    except Exception as e:  # noqa: BLE001
        checks["error"] = repr(e)
    out["checks"] = checks
    out["complete"] = True
    senc.close()
    if penc is not None:
        penc.close()
    return out
