"""Tests of the one-process-per-GPU N>1 path: world_size-2 gloo processes exercising fastecc_amd.sharding.

Without a GPU the encode callable is the ORACLE (tests may use it) — that covers the partitioning, the sub-slab
pipeline, the gather and the re-interleave.  With a GPU (-m gpu) the same two-rank job runs the HIP encoder through the
C ABI (both ranks on device 0; gloo moves the pieces through host memory, where bench.py uses RCCL)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, N, S, sub_slabs, use_gpu, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fastecc_amd import sharding
        from oracle import Oracle
        orc = Oracle()
        rng = np.random.default_rng(123)  # same stripe on every rank
        host = rng.integers(0, 0xFFF00001, size=(N, S), dtype=np.uint64).astype(np.uint32)
        stripe = torch.from_numpy(host.view(np.int32))
        want = orc.encode_fast(host)
        w = S // world

        if use_gpu:
            import fastecc_amd
            enc = fastecc_amd.Encoder(2 * N, N, 4 * w, device=0)
            columns = sharding.hip_columns_encoder(enc)
            my_slab = sharding.take_slab(stripe, rank, world).to("cuda:0")
        else:
            def columns(data_slab, parity_slab, col0, width):  # the oracle on a column range (columns are independent)
                cols = np.ascontiguousarray(data_slab.numpy().view(np.uint32)[:, col0:col0 + width])
                parity_slab[:, col0:col0 + width] = torch.from_numpy(orc.encode_fast(cols).view(np.int32))
            my_slab = sharding.take_slab(stripe, rank, world)

        # (1) one stripe in column slabs: encode + pipelined gather == encode of the whole stripe, on the root only
        pslab, full = sharding.encode_slab_and_gather(my_slab, columns, N, dst=0, sub_slabs=sub_slabs, collective_on_host=use_gpu)
        if use_gpu:
            torch.cuda.synchronize()
        ok_slab = np.array_equal(pslab.cpu().numpy().view(np.uint32), want[:, rank * w:(rank + 1) * w])
        ok_cols = (full is None) if rank != 0 else np.array_equal(full.cpu().numpy().view(np.uint32), want)

        # (1a) the slab resident as contiguous sub-slabs: no pack, the root's part encoded in place, one re-interleave per sub-slab
        H = sharding.sub_slab_count(w, sub_slabs)
        sub = sharding.split_into_sub_slabs(my_slab, H)
        if use_gpu:
            sub_enc = fastecc_amd.Encoder(2 * N, N, 4 * (w // H), device=0)
            def sub_fn(d, o):
                sub_enc.encode(d, o)
        else:
            def sub_fn(d, o):
                o.copy_(torch.from_numpy(orc.encode_fast(np.ascontiguousarray(d.numpy().view(np.uint32))).view(np.int32)))
        for _ in range(2):  # twice: the second call reuses the workspace
            wsp = {}
            mine, full2 = sharding.encode_sub_slabs_and_gather(sub, sub_fn, N, dst=0, collective_on_host=use_gpu, workspace=wsp)
            if use_gpu:
                torch.cuda.synchronize()
            got_mine = mine.permute(1, 0, 2).reshape(N, w).cpu().numpy().view(np.uint32)
            ok_slab = ok_slab and np.array_equal(got_mine, want[:, rank * w:(rank + 1) * w])
            ok_cols = ok_cols and ((full2 is None) if rank != 0 else np.array_equal(full2.cpu().numpy().view(np.uint32), want))
        # (1a') the root's slab lives in full-pitch arrays and its parity is written straight into the full blocks
        if rank == 0:
            full_data = torch.zeros((N, world, H, w // H), dtype=torch.int32, device=my_slab.device)
            full_data[:, 0] = my_slab.view(N, H, w // H)
            sub0 = full_data[:, 0].permute(1, 0, 2)
        else:
            sub0 = sub
        if use_gpu:
            pitched = fastecc_amd.Encoder(2 * N, N, 4 * (w // H), device=0)
            pitched.set_option("row_pitch_words", world * w)
            def sub_fn2(d, o):
                (sub_enc if d.is_contiguous() else pitched).encode(d, o)
        else:
            sub_fn2 = sub_fn
        mine3, full3 = sharding.encode_sub_slabs_and_gather(sub0, sub_fn2, N, dst=0, collective_on_host=use_gpu, workspace={}, root_in_place=True)
        if use_gpu:
            torch.cuda.synchronize()
            pitched.close()
        ok_cols = ok_cols and ((full3 is None) if rank != 0 else np.array_equal(full3.cpu().numpy().view(np.uint32), want))
        if use_gpu:
            sub_enc.close()

        # (1b) the all-gather form: every rank ends with the full parity
        def encode_fn(slab):
            out = torch.empty_like(slab)
            if use_gpu:
                dslab = slab.to("cuda:0")
                dout = torch.empty_like(dslab)
                enc.encode(dslab, dout)
                torch.cuda.synchronize()
                return dout.cpu()
            out.copy_(torch.from_numpy(orc.encode_fast(slab.numpy().view(np.uint32)).view(np.int32)))
            return out
        everywhere = sharding.encode_column_sharded(stripe, encode_fn)
        ok_cols = ok_cols and np.array_equal(everywhere.numpy().view(np.uint32), want)

        # (2) independent stripes: every stripe encoded exactly once across ranks
        mine = sharding.stripes_for_rank(5, rank, world)
        counts = torch.zeros(5, dtype=torch.int64)
        counts[mine] = 1
        dist.all_reduce(counts)
        ok_stripes = bool((counts == 1).all())

        # (3) max-over-ranks timing reduction used by bench.py
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ok_max = float(t.item()) == float(world)
        if use_gpu:
            enc.close()
        q.put((rank, ok_slab and ok_cols, ok_stripes, ok_max))
    finally:
        dist.destroy_process_group()


def _run_two_ranks(N, S, sub_slabs, use_gpu, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, S, sub_slabs, use_gpu, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in results) == list(range(world))
    for rank, ok_cols, ok_stripes, ok_max in results:
        assert ok_cols, "column-sharded encode + gather differs from the full encode (rank %d)" % rank
        assert ok_stripes and ok_max


@pytest.mark.parametrize("N,S,sub_slabs", [(64, 8, 2), (256, 6, 1), (128, 256, 2), (128, 256, 4)])
def test_two_rank_gloo_column_slabs_and_stripes(N, S, sub_slabs):
    _run_two_ranks(N, S, sub_slabs, use_gpu=False)


@pytest.mark.gpu
@pytest.mark.parametrize("N,S,sub_slabs", [(1 << 10, 1024, 2), (1 << 12, 256, 4), (64, 8, 2)])
def test_two_rank_gloo_with_the_hip_encoder(hip_lib, N, S, sub_slabs):
    _run_two_ranks(N, S, sub_slabs, use_gpu=True)


# BASELINE configs[3] IS eight ranks: 4 KB blocks = 1024 words -> 128-word slabs -> two 64-word sub-slabs, k/8 whole blocks per rank.
@pytest.mark.parametrize("N,S,sub_slabs", [(64, 1024, 2), (8, 1024, 2)])
def test_eight_rank_gloo_headline_geometry_gather(N, S, sub_slabs):
    _run_two_ranks(N, S, sub_slabs, use_gpu=False, world=8)


@pytest.mark.gpu
def test_eight_rank_gloo_headline_geometry_gather_with_the_hip_encoder(hip_lib):
    _run_two_ranks(1 << 10, 1024, 2, use_gpu=True, world=8)


@pytest.mark.gpu
def test_single_rank_pipeline_with_the_hip_encoder(hip_lib, oracle):
    """world = 1: the sub-slab pipeline alone (fastecc_encode_columns per sub-slab, pack, unpack) on the device."""
    sys.path.insert(0, ROOT)
    import fastecc_amd
    from fastecc_amd import sharding
    N, w = 1 << 12, 128
    host = np.random.default_rng(3).integers(0, 0xFFF00001, size=(N, w), dtype=np.uint64).astype(np.uint32)
    slab = torch.from_numpy(host.view(np.int32)).to("cuda:0")
    with fastecc_amd.Encoder(2 * N, N, 4 * w, device=0) as enc:
        for sub in (1, 2, 4):
            pslab, full = sharding.encode_slab_and_gather(slab, sharding.hip_columns_encoder(enc), N, sub_slabs=sub)
            torch.cuda.synchronize()
            want = oracle.encode_fast(host)
            assert np.array_equal(pslab.cpu().numpy().view(np.uint32), want)
            assert np.array_equal(full.cpu().numpy().view(np.uint32), want)
    # the sub-slab-resident form: side stream for the re-interleave, device path (no host staging) at world = 1
    for sub in (1, 2, 4):
        with fastecc_amd.Encoder(2 * N, N, 4 * (w // sub), device=0) as enc:
            wsp = {}
            data_sub = sharding.split_into_sub_slabs(slab, sub)
            for _ in range(2):
                mine, full = sharding.encode_sub_slabs_and_gather(data_sub, lambda d, o: enc.encode(d, o, stream=torch.cuda.current_stream().cuda_stream), N, workspace=wsp)
            torch.cuda.synchronize()
            want = oracle.encode_fast(host)
            assert np.array_equal(full.cpu().numpy().view(np.uint32), want)
            assert np.array_equal(mine.permute(1, 0, 2).reshape(N, w).cpu().numpy().view(np.uint32), want)


def _worker_a2a(rank, world, port, N, S, sub_slabs, use_gpu, q):
    """encode_all_to_all: block-distributed parity (and data) against the oracle's encode of the whole stripe."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fastecc_amd import sharding
        from oracle import Oracle
        orc = Oracle()
        host = np.random.default_rng(321).integers(0, 0xFFF00001, size=(N, S), dtype=np.uint64).astype(np.uint32)  # same stripe on every rank
        stripe = torch.from_numpy(host.view(np.int32))
        want = orc.encode_fast(host)
        w = S // world
        H = sharding.sub_slab_count(w, sub_slabs)
        lo, hi = sharding.rows_for_rank(N, rank, world)
        dev = "cuda:0" if use_gpu else "cpu"
        if use_gpu:
            import fastecc_amd
            enc = fastecc_amd.Encoder(2 * N, N, 4 * (w // H), device=0)
            def fn(d, o):
                enc.encode(d, o, stream=torch.cuda.current_stream().cuda_stream)
        else:
            def fn(d, o):
                o.copy_(torch.from_numpy(orc.encode_fast(np.ascontiguousarray(d.numpy().view(np.uint32))).view(np.int32)))
        ok = True
        # (a) data resident as column sub-slabs, parity leaves block-distributed
        sub = sharding.split_into_sub_slabs(sharding.take_slab(stripe, rank, world), H).to(dev)
        wsp = {}
        for _ in range(2):  # the second call reuses the workspace
            mine, blocks = sharding.encode_all_to_all(sub, fn, N, collective_on_host=use_gpu, workspace=wsp)
            if use_gpu:
                torch.cuda.synchronize()
            ok = ok and np.array_equal(blocks.cpu().numpy().view(np.uint32), want[lo:hi])
            ok = ok and np.array_equal(mine.permute(1, 0, 2).reshape(N, w).cpu().numpy().view(np.uint32), want[:, rank * w:(rank + 1) * w])
        # (b) data block-distributed as well: whole data blocks [lo, hi) in, whole parity blocks [lo, hi) out
        my_blocks = stripe[lo:hi].contiguous().to(dev)
        out = torch.full((hi - lo, S), -1, dtype=torch.int32, device=dev)
        _, blocks2 = sharding.encode_all_to_all(my_blocks, fn, N, data_is_blocks=True, sub_slabs=sub_slabs, parity_blocks=out,
                                                collective_on_host=use_gpu, workspace={})
        if use_gpu:
            torch.cuda.synchronize()
            enc.close()
        ok = ok and blocks2 is out and np.array_equal(out.cpu().numpy().view(np.uint32), want[lo:hi])
        # the whole thing reproduces the reference's parity hash when the pieces are put together (checked by the caller)
        q.put((rank, ok, orc.hash(np.ascontiguousarray(out.cpu().numpy().view(np.uint32))) if world == 1 else None))
    finally:
        dist.destroy_process_group()


def _run_a2a(world, N, S, sub_slabs, use_gpu):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_a2a, args=(r, world, port, N, S, sub_slabs, use_gpu, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in results) == list(range(world))
    for rank, ok, _ in results:
        assert ok, "block-distributed encode differs from the full encode (rank %d of %d)" % (rank, world)


@pytest.mark.parametrize("world,N,S,sub_slabs", [(2, 64, 8, 2), (2, 128, 256, 2), (4, 64, 8, 1), (4, 128, 256, 2), (4, 256, 512, 4), (4, 32, 4, 1)])
def test_gloo_all_to_all_block_distributed(world, N, S, sub_slabs):
    _run_a2a(world, N, S, sub_slabs, use_gpu=False)


# eight ranks at the headline geometry (128-word slabs in two 64-word sub-slabs), data in slabs and block-distributed (both in _worker_a2a)
@pytest.mark.parametrize("N,S,sub_slabs", [(64, 1024, 2), (8, 1024, 2), (128, 1024, 4)])
def test_eight_rank_gloo_all_to_all_headline_geometry(N, S, sub_slabs):
    from fastecc_amd import sharding
    assert sharding.sub_slab_count(1024 // 8, 2) == 2 and 1024 // 8 // 2 == 64
    _run_a2a(8, N, S, sub_slabs, use_gpu=False)


@pytest.mark.gpu
def test_eight_rank_gloo_all_to_all_headline_geometry_with_the_hip_encoder(hip_lib):
    _run_a2a(8, 1 << 11, 1024, 2, use_gpu=True)


@pytest.mark.gpu
@pytest.mark.parametrize("world,N,S,sub_slabs", [(2, 1 << 10, 1024, 2), (4, 1 << 12, 1024, 2), (4, 64, 256, 4)])
def test_gloo_all_to_all_with_the_hip_encoder(hip_lib, world, N, S, sub_slabs):
    _run_a2a(world, N, S, sub_slabs, use_gpu=True)


def test_slab_helpers_roundtrip():
    sys.path.insert(0, ROOT)
    from fastecc_amd import sharding
    x = torch.arange(4 * 12, dtype=torch.int32).reshape(4, 12)
    for world in (1, 2, 3, 4):
        slabs = [sharding.take_slab(x, r, world) for r in range(world)]
        assert all(s.shape == (4, 12 // world) for s in slabs)
        assert torch.equal(sharding.merge_slabs(slabs), x)
    with pytest.raises(ValueError):
        sharding.slab_bounds(10, 0, 4)
    assert sharding.stripes_for_rank(7, 1, 3) == [1, 4]
    assert [sharding.sub_slab_count(w, 4) for w in (128, 64, 32, 96, 8)] == [4, 2, 1, 1, 1]
    assert sharding.sub_slab_count(128, 1) == 1 and sharding.sub_slab_count(128, 3) == 2
