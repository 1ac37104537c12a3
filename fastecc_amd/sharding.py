"""Multi-GPU host logic for the encode path (one process per GPU, torch.distributed over RCCL).

The encode has NO exchange step: the S word columns of a stripe are independent length-N transforms
(ntt.cpp:348-350), and different stripes are independent jobs.  Two ways to use N GPUs:

  * independent stripes  — rank r encodes stripes r, r+world, ...; nothing is communicated.  This is what
    bench.py times (weak scaling, no collective in the timed region).
  * column slabs of ONE stripe — rank r encodes words [r*S/world, (r+1)*S/world) of every block with an
    encoder built for block_bytes/world, then the slabs are all-gathered and re-interleaved.  The gather
    is pure data movement (xGMI), reported separately by bench.py --gather.

Everything here is index arithmetic on torch tensors; the encode itself is passed in as a callable so the
same code runs with the HIP encoder on GPUs and — in the CPU unit tests only — with the oracle.
"""
import torch


def stripes_for_rank(n_stripes, rank, world):
    """Round-robin assignment of independent stripes to ranks."""
    return list(range(rank, n_stripes, world))


def slab_bounds(words_per_block, rank, world):
    """Word range [lo, hi) of a block owned by `rank`; the block must split evenly."""
    if words_per_block % world:
        raise ValueError("words_per_block=%d is not divisible by world=%d" % (words_per_block, world))
    w = words_per_block // world
    return rank * w, (rank + 1) * w


def take_slab(stripe, rank, world):
    """stripe: [N, S] int32 tensor (block-major) -> contiguous [N, S/world] slab of this rank."""
    lo, hi = slab_bounds(stripe.shape[1], rank, world)
    return stripe[:, lo:hi].contiguous()


def merge_slabs(slabs):
    """Inverse of take_slab over all ranks: list of [N, S/world] -> [N, S]."""
    return torch.cat(list(slabs), dim=1).contiguous()


def encode_column_sharded(stripe, encode_fn, group=None):
    """Encode one stripe cooperatively: every rank holds the full `stripe` ([N, S] int32), encodes its
    column slab with `encode_fn(slab) -> parity_slab`, and all ranks end with the full parity stripe.

    encode_fn must be an encoder for blocks of S/world words (e.g. fastecc_amd.Encoder(2N, N, 4*S/world))."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    mine = encode_fn(take_slab(stripe, rank, world))
    if world == 1:
        return mine
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine, group=group)
    return merge_slabs(gathered)
