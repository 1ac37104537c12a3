"""Multi-GPU host logic for the encode path when the host runs ONE PROCESS PER GPU (torch.distributed over RCCL).

The encode has no exchange step: the S word columns of a stripe are independent length-N transforms
(ntt.cpp:348-350), and different stripes are independent jobs.  Two ways to use G GPUs:

  * independent stripes ("replicas") — rank r encodes stripes r, r+G, ...; nothing is communicated;
  * ONE stripe in column slabs (BASELINE.json configs[3]) — rank r owns words [r*S/G, (r+1)*S/G) of every block
    (a [k, S/G] slab resident in its HBM), encodes them with an encoder for block_bytes/G, and the parity slabs are
    gathered over xGMI into full 4 KB parity blocks on one rank: `encode_sub_slabs_and_gather` (slab resident as contiguous
    column sub-slabs: no pack, the root's own part never moves, one re-interleaving kernel per sub-slab, transfers and
    re-interleave on a side stream under the next sub-slab's encode) or `encode_slab_and_gather` (slab resident as one
    [k, w] array: fastecc_encode_columns per sub-slab, a pack per sub-slab, a copy per received piece).  Gather-to-root is
    bound by the root's links ((G-1)/G of the stripe enters one GPU); `encode_all_to_all` leaves the result BLOCK-DISTRIBUTED
    instead — rank g ends with parity blocks [g*M/G, (g+1)*M/G) whole, every link carries 1/G^2 of the stripe — and can take
    block-distributed data as well (the mirror transpose in front of the encode).

The single-process form of the same thing (one host thread driving all GPUs, peer copies instead of RCCL) is
fastecc_create_sharded / fastecc_encode_sharded in the C ABI (csrc/sharded.hip).

Everything here is index arithmetic on torch tensors plus collectives; the encode itself is passed in as a callable
so the same code runs with the HIP encoder on GPUs and — in the CPU unit tests only — with the oracle.
"""
import os

import torch

# Test hook (tests/test_gpu_rccl_single_rank.py): with a ONE-rank process group the collectives are normally skipped; set, they are issued anyway,
# so that a 1-GPU box runs the very RCCL calls of the N > 1 path (arguments, dtypes, contiguity, streams) against the real library.
FORCE_COLLECTIVES = bool(os.environ.get("FASTECC_SHARDING_FORCE_COLLECTIVES"))
if FORCE_COLLECTIVES:  # a stray exported variable must not change production timings silently
    import sys
    print("[fastecc_amd.sharding] TEST HOOK ACTIVE: FASTECC_SHARDING_FORCE_COLLECTIVES is set - one-rank groups issue their collectives too "
          "(timings are NOT those of a production run)", file=sys.stderr, flush=True)


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


def sub_slab_count(slab_words, wanted, words_per_column=1):
    """Largest power of two <= wanted that cuts the slab into sub-slabs of whole 32-word (128-byte) row segments.
    slab_words counts tensor columns of words_per_column 4-byte words each."""
    h = 1
    while h * 2 <= wanted and (slab_words * words_per_column) % (32 * h * 2) == 0:
        h *= 2
    return h


def hip_columns_encoder(enc, words_per_column=1):
    """encode_columns callable for encode_slab_and_gather from a fastecc_amd.Encoder built for the slab's block size.
    words_per_column: 4-byte words per tensor column (1 for int32 slabs, 2 for the int64 slabs of the 64-bit field)."""
    def fn(data_slab, parity_slab, col0, width):
        stream = torch.cuda.current_stream(data_slab.device).cuda_stream if data_slab.is_cuda else 0
        if col0 == 0 and width == data_slab.shape[1]:
            enc.encode(data_slab, parity_slab, stream=stream)
        else:
            enc.encode_columns(data_slab, parity_slab, col0 * words_per_column, width * words_per_column, stream=stream)
    return fn


def encode_slab_and_gather(data_slab, encode_columns, parity_rows, parity_full=None, dst=0, sub_slabs=2, group=None,
                           collective_on_host=False, workspace=None):
    """Encode this rank's column slab and gather all ranks' parity slabs into full parity blocks on rank `dst`.

    data_slab      [k, w] int32, contiguous, on this rank's device: words [rank*w, (rank+1)*w) of the k data blocks.
    encode_columns callable(data_slab, parity_slab, col0, width): parity_slab[:, col0:col0+width] <- encode of those
                   columns (hip_columns_encoder(...) on GPUs).
    parity_rows    n - k.
    parity_full    [n-k, world*w] int32 on rank `dst` (allocated if None there); ignored elsewhere.
    sub_slabs      pipeline depth: the slab is cut into this many column sub-slabs (rounded down to what splits
                   into whole 128-byte row segments).
    collective_on_host  stage the collective's buffers through host memory (CPU tests of the GPU path with gloo).
    workspace      optional dict: the parity slab, send and receive buffers are kept in it and reused by later calls
                   (a steady-state caller allocates nothing per stripe).

    Returns (parity_slab, parity_full or None).  Order of work per sub-slab h:
        encode h -> pack h (strided [n-k, ws] -> contiguous) -> async gather h to dst   | overlaps encode h+1
        dst: wait gather h -> write the G received pieces into parity_full[:, g*w + h*ws : ...]
    """
    import torch.distributed as dist
    ranked = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if ranked else 1
    rank = dist.get_rank(group) if ranked else 0
    k, w = data_slab.shape
    H = sub_slab_count(w, sub_slabs, data_slab.element_size() // 4)
    ws = w // H
    dev = data_slab.device
    ws_ = workspace if workspace is not None else {}
    key = (parity_rows, w, H, world, str(dev), collective_on_host)
    if ws_.get("key") != key:
        ws_.clear()
        ws_["key"] = key
    cdev = torch.device("cpu") if collective_on_host else dev

    def buf(name, shape, device):
        t = ws_.get(name)
        if t is None:
            t = ws_[name] = torch.empty(shape, dtype=data_slab.dtype, device=device)
        return t

    parity_slab = buf("parity_slab", (parity_rows, w), dev)
    if rank == dst and parity_full is None:
        parity_full = buf("parity_full", (parity_rows, world * w), dev)
    pending = []
    for h in range(H):
        encode_columns(data_slab, parity_slab, h * ws, ws)
        piece = parity_slab[:, h * ws:(h + 1) * ws]
        if H > 1:                                               # pack: RCCL moves contiguous buffers
            send = buf("send%d" % h, (parity_rows, ws), dev)
            send.copy_(piece)
        else:
            send = piece
        if collective_on_host:
            send = send.cpu()
        if world == 1 and not (FORCE_COLLECTIVES and ranked):
            pending.append((h, None, [send]))
            continue
        recv = [buf("recv%d_%d" % (h, g), (parity_rows, ws), cdev) for g in range(world)] if rank == dst else None
        work = dist.gather(send, gather_list=recv, dst=dst, group=group, async_op=True)
        pending.append((h, work, recv))
        # the root re-interleaves sub-slab h-1 while sub-slab h is on the wire
        if len(pending) >= 2:
            _unpack(pending.pop(0), parity_full, rank, dst, w, ws)
    while pending:
        _unpack(pending.pop(0), parity_full, rank, dst, w, ws)
    return parity_slab, (parity_full if rank == dst else None)


def _unpack(item, parity_full, rank, dst, w, ws):
    h, work, recv = item
    if work is not None:
        work.wait()  # device collectives: the current stream waits; host collectives: blocks
    if rank != dst or recv is None:
        return
    for g, piece in enumerate(recv):
        parity_full[:, g * w + h * ws: g * w + (h + 1) * ws].copy_(piece, non_blocking=True)


def split_into_sub_slabs(slab, sub_slabs):
    """[rows, w] column slab -> [H, rows, w/H] contiguous sub-slabs (the residency encode_sub_slabs_and_gather works on)."""
    rows, w = slab.shape
    return slab.view(rows, sub_slabs, w // sub_slabs).permute(1, 0, 2).contiguous()


def encode_sub_slabs_and_gather(data_sub, encode_fn, parity_rows, parity_full=None, dst=0, group=None, collective_on_host=False, workspace=None,
                                root_in_place=False):
    """The gather of BASELINE configs[3] without a pack and with ONE re-interleaving kernel per sub-slab on the root.

    A rank's slab is resident as H contiguous SUB-SLABS (data_sub: [H, k, ws], words [rank*w + h*ws, ...) of every block, w = H*ws),
    which is how a scatter would deliver them anyway.  Each sub-slab is an ordinary contiguous stripe of narrower blocks:

        encode_fn(data_sub[h], out[rows, ws])     every rank, on the current stream; the root's `out` is its own slot of the receive
                                                  buffer, so its contribution never travels and is never copied
        gather(out -> recv[h][g])                 RCCL (point-to-point sends over xGMI: every peer uses its own link to the root),
                                                  on a side stream, so sub-slab h travels while sub-slab h+1 is being encoded
        parity_full[:, g*w + h*ws ...] <- recv    root: one strided kernel for all `world` pieces of the sub-slab, on the side
                                                  stream as well (it overlaps the next encode and the next transfer)

    root_in_place: the root's encode_fn accepts row-strided tensors (an encoder with the "row_pitch_words" option set to the full block), its
    data sub-slabs are views into a [k, world*w] array, and its parity is written straight into its columns of parity_full — the root's part
    is then neither gathered nor re-interleaved (at world = 1 the call is the encode itself).

    Returns (this rank's parity sub-slabs [H, rows, ws], parity_full [rows, world*w] on the root or None).  workspace: dict keeping
    the buffers and the side stream between calls.  collective_on_host: the CPU / gloo form of the same control flow (tests).
    """
    import torch.distributed as dist
    ranked = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if ranked else 1
    rank = dist.get_rank(group) if ranked else 0
    H, k, ws = data_sub.shape
    w = H * ws
    dev = data_sub.device
    ws_ = workspace if workspace is not None else {}
    key = ("sub", parity_rows, ws, H, world, str(dev), collective_on_host)
    if ws_.get("key") != key:
        ws_.clear()
        ws_["key"] = key
    root = rank == dst

    def buf(name, shape, device=dev):
        t = ws_.get(name)
        if t is None:
            t = ws_[name] = torch.empty(shape, dtype=data_sub.dtype, device=device)
        return t

    # root: [H][world][rows][ws], its own results land in [:, rank]; other ranks: [H][rows][ws]
    recv = buf("recv", (H, world, parity_rows, ws)) if root else None
    if root and parity_full is None:
        parity_full = buf("parity_full", (parity_rows, world * w))
    full4 = parity_full.view(parity_rows, world, H, ws) if root else None
    in_place = bool(root_in_place and root)
    if in_place:
        mine = full4[:, rank].permute(1, 0, 2)  # [H, rows, ws] views into the full blocks
        others = [g for g in range(world) if g != rank]
    else:
        mine = recv[:, rank] if root else buf("send", (H, parity_rows, ws))
    on_gpu = dev.type == "cuda" and not collective_on_host
    if on_gpu:
        side = ws_.get("side_stream")
        if side is None:
            side = ws_["side_stream"] = torch.cuda.Stream(device=dev)
        main = torch.cuda.current_stream(dev)
    for h in range(H):
        encode_fn(data_sub[h], mine[h])
        if on_gpu:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                if world > 1 or (FORCE_COLLECTIVES and ranked):
                    # in place: the root's part is already home; its slot of the gather gets a scratch tensor (distinct memory from the
                    # receive slot, so that no backend has to cope with an input that aliases its own output)
                    piece = buf("root_dummy", (parity_rows, ws)) if in_place else mine[h]
                    dist.gather(piece, gather_list=[recv[h, g] for g in range(world)] if root else None, dst=dst, group=group, async_op=True).wait()
                if root and not in_place:
                    full4[:, :, h, :].copy_(recv[h].permute(1, 0, 2), non_blocking=True)
                elif root and others:
                    idx = ws_.get("others_idx")
                    if idx is None:
                        idx = ws_["others_idx"] = torch.tensor(others, device=dev)
                    full4[:, :, h, :].index_copy_(1, idx, recv[h].index_select(0, idx).permute(1, 0, 2))
        else:
            if world > 1:
                piece = mine[h].contiguous().cpu() if collective_on_host else mine[h].contiguous()
                got = [torch.empty_like(piece) for _ in range(world)] if root else None
                dist.gather(piece, gather_list=got, dst=dst, group=group)
                if root:
                    for g in range(world):
                        if not (in_place and g == rank):
                            recv[h, g].copy_(got[g])
            if root and not in_place:
                full4[:, :, h, :].copy_(recv[h].permute(1, 0, 2))
            elif root:
                for g in others:
                    full4[:, g, h, :].copy_(recv[h, g])
    if on_gpu:
        main.wait_stream(side)  # the call behaves as one operation on the caller's stream
    return mine, (parity_full if root else None)


def rows_for_rank(rows, rank, world):
    """Block range [lo, hi) that `rank` holds WHOLE in the block-distributed layout; the blocks must split evenly."""
    if rows % world:
        raise ValueError("%d blocks do not divide over %d ranks" % (rows, world))
    return rank * (rows // world), (rank + 1) * (rows // world)


def encode_all_to_all(data, encode_fn, parity_rows, data_is_blocks=False, sub_slabs=2, parity_blocks=None, group=None, collective_on_host=False,
                      workspace=None):
    """ONE stripe over the ranks with a BLOCK-DISTRIBUTED result (and, optionally, input): the N data and M parity blocks of RS.md:13-33 are
    separately stored units, so after the encode rank g holds parity blocks [g*M/G, (g+1)*M/G) WHOLE — what a host that disperses blocks to
    devices wants — instead of everything converging on one root.

    The compute is the column-slab encode (no exchange inside the transform); the exchange is an all-to-all in which every rank sends 1/G of its
    slab to every other rank, so every xGMI link carries 1/G^2 of the stripe per direction and no rank is a hot spot (gather-to-root puts
    (G-1)/G of the stripe into ONE rank's links):

        parity sub-slab h of rank g, [M, ws]      rows are block indices: the G row chunks are contiguous, so all_to_all_single sends them as
                                                  they lie — no pack
        recv [G, M/G, ws]                         chunk j = rank j's columns of MY parity blocks
        parity_blocks[:, j, h, :] <- recv[j]      one strided kernel per sub-slab (on the side stream, under the next sub-slab's encode)

    data_is_blocks: the input is block-distributed as well — `data` = this rank's whole data blocks [k/G, S] (blocks [g*k/G, (g+1)*k/G)).  The
    mirror transpose runs first, per sub-slab: pack [k/G, G, ws] -> [G, k/G, ws] (one strided kernel), all_to_all_single, and what arrives,
    [G, k/G, ws], IS the contiguous [k, ws] data sub-slab (no unpack).  Sub-slab h+1 is transposed while sub-slab h is encoded.
    Otherwise `data` is [H, k, ws]: the slab resident as H contiguous column sub-slabs (encode_sub_slabs_and_gather's residency).

    encode_fn(data_sub_slab [k, ws], out [M, ws]) as in encode_sub_slabs_and_gather.  Returns (parity sub-slabs [H, M, ws] of this rank's
    columns, parity_blocks [M/G, G*H*ws] = this rank's whole parity blocks).  The call behaves as one operation on the caller's stream.
    """
    import torch.distributed as dist
    ranked = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if ranked else 1
    rank = dist.get_rank(group) if ranked else 0
    dev = data.device
    if data_is_blocks:
        kg, S = data.shape
        k = kg * world
        if S % world:
            raise ValueError("block of %d words does not split over %d ranks" % (S, world))
        w = S // world
        H = sub_slab_count(w, sub_slabs, data.element_size() // 4)
        ws = w // H
    else:
        H, k, ws = data.shape
        w = H * ws
        if k % world:
            raise ValueError("%d data blocks do not divide over %d ranks" % (k, world))
        kg = k // world
    if parity_rows % world:
        raise ValueError("%d parity blocks do not divide over %d ranks" % (parity_rows, world))
    mg = parity_rows // world
    ws_ = workspace if workspace is not None else {}
    key = ("a2a", k, parity_rows, ws, H, world, str(dev), collective_on_host, bool(data_is_blocks))
    if ws_.get("key") != key:
        ws_.clear()
        ws_["key"] = key

    def buf(name, shape):
        t = ws_.get(name)
        if t is None:
            t = ws_[name] = torch.empty(shape, dtype=data.dtype, device=dev)
        return t

    mine = buf("parity_sub", (H, parity_rows, ws))
    recv = buf("recv", (H, world, mg, ws))
    if parity_blocks is None:
        parity_blocks = buf("parity_blocks", (mg, world * w))
    out4 = parity_blocks.view(mg, world, H, ws)
    if data_is_blocks:
        data_sub = buf("data_sub", (H, k, ws))
        send_in = buf("send_in", (H, world, kg, ws))
        in4 = data.view(kg, world, H, ws)
    else:
        data_sub = data
    on_gpu = dev.type == "cuda" and not collective_on_host

    def exchange(dst, src):
        """dst[j] <- rank j's src[rank] (dst, src: [world, rows, ws] contiguous)."""
        if world == 1 and not (FORCE_COLLECTIVES and ranked):
            dst.copy_(src)
        elif collective_on_host:
            got = torch.empty(src.shape, dtype=src.dtype)
            dist.all_to_all_single(got, src.cpu(), group=group)
            dst.copy_(got)
        else:
            dist.all_to_all_single(dst, src, group=group)

    def transpose_in(h):
        send_in[h].copy_(in4[:, :, h, :].permute(1, 0, 2))                          # pack: [kg, G, ws] -> [G, kg, ws]
        exchange(data_sub[h].view(world, kg, ws), send_in[h])                       # arrives as the contiguous [k, ws] sub-slab

    def transpose_out(h):
        exchange(recv[h], mine[h].view(world, mg, ws))                              # row chunks of the parity sub-slab travel as they lie
        out4[:, :, h, :].copy_(recv[h].permute(1, 0, 2))                            # re-interleave into whole blocks

    if on_gpu:
        side_in, side_out = ws_.get("side_in"), ws_.get("side_out")
        if side_in is None:
            side_in = ws_["side_in"] = torch.cuda.Stream(device=dev)
            side_out = ws_["side_out"] = torch.cuda.Stream(device=dev)
        main = torch.cuda.current_stream(dev)
        arrived = [None] * H
        if data_is_blocks:
            side_in.wait_stream(main)
            with torch.cuda.stream(side_in):
                transpose_in(0)
                arrived[0] = side_in.record_event()
        for h in range(H):
            if data_is_blocks:
                if h + 1 < H:                                                       # the next sub-slab's input travels under this encode
                    with torch.cuda.stream(side_in):
                        transpose_in(h + 1)
                        arrived[h + 1] = side_in.record_event()
                main.wait_event(arrived[h])
            encode_fn(data_sub[h], mine[h])
            side_out.wait_stream(main)
            with torch.cuda.stream(side_out):
                transpose_out(h)
        main.wait_stream(side_out)
        if data_is_blocks:
            main.wait_stream(side_in)
    else:
        for h in range(H):
            if data_is_blocks:
                transpose_in(h)
            encode_fn(data_sub[h], mine[h])
            transpose_out(h)
    return mine, parity_blocks


def encode_column_sharded(stripe, encode_fn, group=None):
    """Every rank holds the full `stripe` ([N, S] int32), encodes its column slab with `encode_fn(slab) -> parity_slab`,
    and ALL ranks end with the full parity stripe (all_gather; the unpipelined form, kept for hosts that want the parity
    everywhere)."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    mine = encode_fn(take_slab(stripe, rank, world))
    if world == 1:
        return mine
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine, group=group)
    return merge_slabs(gathered)
