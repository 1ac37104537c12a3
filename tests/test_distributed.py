"""CPU tests of the N>1 path: world_size-2 gloo processes exercising fastecc_amd.sharding.

The encode callable is the ORACLE here (tests may use it); on GPUs bench.py passes the HIP encoder through
the same functions."""
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


def _worker(rank, world, port, N, S, q):
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

        def encode_fn(slab):
            out = orc.encode_fast(slab.numpy().view(np.uint32))
            return torch.from_numpy(out.view(np.int32))

        # (1) column slabs of one stripe + all_gather == encode of the whole stripe
        full = sharding.encode_column_sharded(stripe, encode_fn)
        want = orc.encode_fast(host)
        ok_cols = np.array_equal(full.numpy().view(np.uint32), want)

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
        q.put((rank, ok_cols, ok_stripes, ok_max))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("N,S", [(64, 8), (256, 6)])
def test_two_rank_gloo_column_slabs_and_stripes(N, S):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, S, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in results) == [0, 1]
    for rank, ok_cols, ok_stripes, ok_max in results:
        assert ok_cols, "column-sharded encode + all_gather differs from the full encode (rank %d)" % rank
        assert ok_stripes and ok_max


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
