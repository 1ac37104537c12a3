"""The RCCL calls of the N > 1 path against the real library, on ONE GPU.

RCCL refuses two ranks on one device, so a 1-GPU box cannot run the multi-rank job itself (the gloo tests cover its control flow).  What it CAN run
is a one-rank `nccl` process group with the collectives forced on (FASTECC_SHARDING_FORCE_COLLECTIVES): the very `all_to_all_single` / `gather` /
`new_group` / `barrier` / `all_reduce` calls, with the tensors, dtypes, views and side streams fastecc_amd/sharding.py and bench.py hand them.
Results are checked against the oracle.  A child process per case (the process group and the env hook are process-wide)."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

CHILD = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["FASTECC_ROOT"])
import fastecc_amd
from fastecc_amd import sharding
from oracle import Oracle

assert sharding.FORCE_COLLECTIVES
print("imports ok", flush=True)
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
dist.barrier()
print("RCCL group up", flush=True)
t = torch.tensor([3.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t.item()) == 3.5
got = [None]
dist.all_gather_object(got, {"rank": 0})
assert got == [{"rank": 0}]
own = dist.new_group(backend="nccl")            # bench.py: the modes after `value` get communicators of their own

orc = Oracle()
N, S, H = 1 << 11, 1024, 2
host = np.random.default_rng(5).integers(0, 0xFFF00001, size=(N, S), dtype=np.uint64).astype(np.uint32)
want = orc.encode_fast(host)
stripe = torch.from_numpy(host.view(np.int32)).to(dev)
enc = fastecc_amd.Encoder(2 * N, N, 4 * (S // H), device=0)
def fn(d, o):
    enc.encode(d, o, stream=torch.cuda.current_stream().cuda_stream)
sub = sharding.split_into_sub_slabs(stripe, H)
for group in (None, own):
    # block-distributed parity (all_to_all_single on a side stream under the next sub-slab's encode), data in slabs and block-distributed
    wsp = {}
    for _ in range(2):
        mine, blocks = sharding.encode_all_to_all(sub, fn, N, workspace=wsp, group=group)
        torch.cuda.synchronize()
        assert np.array_equal(blocks.cpu().numpy().view(np.uint32), want)
        assert np.array_equal(mine.permute(1, 0, 2).reshape(N, S).cpu().numpy().view(np.uint32), want)
    _, blocks2 = sharding.encode_all_to_all(stripe, fn, N, data_is_blocks=True, sub_slabs=H, workspace={}, group=group)
    torch.cuda.synchronize()
    assert np.array_equal(blocks2.cpu().numpy().view(np.uint32), want)
    # gather to the root: plain, and with the root's part encoded in place at the full block pitch
    mine, full = sharding.encode_sub_slabs_and_gather(sub, fn, N, dst=0, workspace={}, group=group)
    torch.cuda.synchronize()
    assert np.array_equal(full.cpu().numpy().view(np.uint32), want)
    pslab, full2 = sharding.encode_slab_and_gather(stripe, sharding.hip_columns_encoder(fastecc_amd.Encoder(2 * N, N, 4 * S, device=0)), N, sub_slabs=H, group=group)
    torch.cuda.synchronize()
    assert np.array_equal(full2.cpu().numpy().view(np.uint32), want)
# the 64-bit field's tensors (int64) through the same calls
from oracle import OracleP61
o61 = OracleP61()
N6, elems = 256, 64
x6 = o61.fill_splitmix(N6, elems, 9)
enc6 = fastecc_amd.Encoder(2 * N6, N6, 16 * (elems // 2), device=0, field=fastecc_amd.FIELD_GF_P61_SQUARED)
def fn6(d, o):
    enc6.encode(d, o, stream=torch.cuda.current_stream().cuda_stream)
s6 = torch.from_numpy(x6.view(np.int64)).to(dev)
sub6 = sharding.split_into_sub_slabs(s6, 2)
_, b6 = sharding.encode_all_to_all(sub6, fn6, N6, workspace={})
torch.cuda.synchronize()
assert np.array_equal(b6.cpu().numpy().view(np.uint64), o61.encode(x6))
dist.barrier()
dist.destroy_process_group()
print("RCCL single-rank ok", torch.cuda.nccl.version())
'''


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_sharding_collectives_on_real_rccl_with_one_rank(hip_lib):
    env = dict(os.environ, FASTECC_SHARDING_FORCE_COLLECTIVES="1", FASTECC_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    if "imports ok" in r.stdout and "RCCL group up" not in r.stdout:  # the library itself did not come up on this box (no fabric / shared memory for it): nothing of ours ran
        pytest.skip("RCCL could not initialise a one-rank group here: " + (r.stderr or r.stdout)[-400:])
    assert r.returncode == 0 and "RCCL single-rank ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
