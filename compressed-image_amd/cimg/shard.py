"""Sharding of independent chunks / channels over ranks.

Chunks are self-contained blosc2 buffers, so the codec path needs no data-path collective: every rank
compresses / decompresses its own items with its own engine.  The small exchanges are the
max-over-ranks of the elapsed time and the per-item compressed sizes (every rank learns the global layout); the one
payload exchange -- finished chunks to the rank that wants the whole image -- is gather_chunks below.  Works with any torch.distributed backend ("nccl" = RCCL on the GPU box, "gloo"
in the CPU tests).
"""
import numpy as np


def partition(n_items, world, rank, items_per_group=1):
    """Items owned by `rank`: round-robin over groups of `items_per_group` consecutive items.

    items_per_group = chunks per channel gives the channel-granular round-robin of SURVEY.md section 8e
    (config 4: 256 channels over 8 GPUs); items_per_group = 1 is the chunk-granular fallback used when
    there are fewer channels than ranks (config 2 at 8 GPUs: 32 chunks, 4 per GPU).
    """
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank / world")
    n_groups = -(-n_items // items_per_group)
    if n_groups < world and items_per_group > 1:
        return partition(n_items, world, rank, 1)
    idx = []
    for g in range(rank, n_groups, world):
        idx.extend(range(g * items_per_group, min((g + 1) * items_per_group, n_items)))
    return np.asarray(idx, dtype=np.int64)


def gather_sizes(dist, local_idx, local_sizes, n_items, device="cpu"):
    """All ranks learn every item's compressed size (int64 array of n_items). One all_reduce(SUM)."""
    import torch
    full = torch.zeros(n_items, dtype=torch.int64, device=device)
    if len(local_idx):
        full[torch.as_tensor(np.asarray(local_idx), dtype=torch.int64, device=device)] = torch.as_tensor(
            np.asarray(local_sizes, dtype=np.int64), device=device)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(full, op=dist.ReduceOp.SUM)
    return full.cpu().numpy()


def max_over_ranks(dist, seconds, device="cpu"):
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_chunks(dist, world, rank, local_idx, local_buf, local_off, sizes_all, n_items, dst=0, items_per_group=1,
                  device="cpu", as_tensor=False):
    """The one real exchange step of SURVEY.md section 8e: owner ranks send their finished chunks to `dst`.

    local_buf   uint8 torch tensor (or numpy array) holding this rank's chunks, chunk k of `local_idx` at
                local_off[k] with length sizes_all[local_idx[k]]
    sizes_all   result of gather_sizes() (every rank knows every chunk's compressed size)
    Every rank packs its chunks back to back (compressed bytes only) and the exchange is point to point with EXACT
    sizes: `dst` posts one receive per owner rank, every other rank one send (batch_isend_irecv; with backend "nccl"
    that is one RCCL group of send/recv over the direct xGMI links -- all-to-one, no ring, nothing padded to the largest
    rank).  Returns on `dst` the chunks in global order -- a list of uint8 numpy arrays, or with as_tensor=True the
    tuple (packed uint8 tensor on `device`, int64 offsets[n_items], int64 sizes[n_items]) -- and None elsewhere.
    """
    import torch
    sizes_all = np.asarray(sizes_all, dtype=np.int64)
    owners = [partition(n_items, world, r, items_per_group) for r in range(world)]
    totals = [int(sizes_all[o].sum()) for o in owners]
    if not torch.is_tensor(local_buf):
        local_buf = torch.as_tensor(np.ascontiguousarray(local_buf).view(np.uint8))
    local_idx = np.asarray(local_idx, dtype=np.int64)
    pieces = [local_buf[int(local_off[k]):int(local_off[k]) + int(sizes_all[g])] for k, g in enumerate(local_idx)]
    packed = (torch.cat(pieces) if pieces else torch.zeros(0, dtype=torch.uint8)).to(device)
    assert packed.numel() == totals[rank]
    multi = dist is not None and dist.is_initialized() and world > 1
    parts = None
    if rank == dst:
        parts = [packed if r == dst else torch.empty(totals[r], dtype=torch.uint8, device=device) for r in range(world)]
    if multi:
        ops = []
        if rank == dst:
            ops = [dist.P2POp(dist.irecv, parts[r], r) for r in range(world) if r != dst and totals[r] > 0]
        elif totals[rank] > 0:
            ops = [dist.P2POp(dist.isend, packed, dst)]
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
    if rank != dst:
        return None
    # global order: chunk g lives in its owner's part at the running offset of that part
    part_base = np.concatenate([[0], np.cumsum(totals[:-1])]).astype(np.int64)
    offsets = np.zeros(n_items, dtype=np.int64)
    for r in range(world):
        at = int(part_base[r])
        for g in owners[r]:
            offsets[int(g)] = at
            at += int(sizes_all[g])
    whole = torch.cat(parts) if world > 1 else parts[0]
    if as_tensor:
        return whole, offsets, sizes_all.copy()
    host = whole.cpu().numpy()
    return [host[int(offsets[g]):int(offsets[g]) + int(sizes_all[g])].copy() for g in range(n_items)]


def scatter_chunks(dist, world, rank, whole, offsets, sizes_all, n_items, src=0, items_per_group=1, device="cpu"):
    """The decode mirror of gather_chunks ("decode is the mirror -- scatter compressed"): rank `src` holds every
    chunk of an image (packed or not: chunk g at offsets[g], sizes_all[g] bytes of `whole`) and every owner receives ITS chunks,
    back to back, to decompress them with its own engine.

    sizes_all   int64[n_items], known to every rank (the table of compressed sizes travels first: broadcast_sizes below, or the
                result of gather_sizes when the chunks were compressed by the same job)
    whole / offsets are only read on `src`.  Point to point with exact sizes, as in gather_chunks: `src` posts one send per
    owner rank, every other rank one receive (one RCCL group over the direct xGMI links under "nccl"; one-to-all, nothing padded).
    Returns on every rank (packed uint8 tensor on `device` holding this rank's chunks, int64 local offsets[len(mine)], the global
    indices `mine`): chunk mine[k] lies at local_offsets[k], sizes_all[mine[k]] bytes.
    """
    import torch
    sizes_all = np.asarray(sizes_all, dtype=np.int64)
    owners = [partition(n_items, world, r, items_per_group) for r in range(world)]
    totals = [int(sizes_all[o].sum()) for o in owners]
    mine = owners[rank]
    local_off = np.concatenate([[0], np.cumsum(sizes_all[mine][:-1])]).astype(np.int64) if len(mine) else np.zeros(0, np.int64)
    multi = dist is not None and dist.is_initialized() and world > 1
    packs = None
    if rank == src:
        if not torch.is_tensor(whole):
            whole = torch.as_tensor(np.ascontiguousarray(whole).view(np.uint8))
        offsets = np.asarray(offsets, dtype=np.int64)
        packs = []
        for r in range(world):
            pieces = [whole[int(offsets[g]):int(offsets[g]) + int(sizes_all[g])] for g in owners[r]]
            packs.append((torch.cat(pieces) if pieces else torch.zeros(0, dtype=torch.uint8)).to(device))
        got = packs[src]
    else:
        got = torch.empty(totals[rank], dtype=torch.uint8, device=device)
    if multi:
        ops = []
        if rank == src:
            ops = [dist.P2POp(dist.isend, packs[r], r) for r in range(world) if r != src and totals[r] > 0]
        elif totals[rank] > 0:
            ops = [dist.P2POp(dist.irecv, got, src)]
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
    assert got.numel() == totals[rank]
    return got, local_off, mine


def broadcast_sizes(dist, sizes, n_items, src=0, device="cpu"):
    """Every rank learns the compressed size of every chunk that rank `src` holds (int64[n_items]): one broadcast."""
    import torch
    t = torch.zeros(n_items, dtype=torch.int64, device=device)
    if sizes is not None:
        t[:] = torch.as_tensor(np.asarray(sizes, dtype=np.int64), device=device)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(t, src=src)
    return t.cpu().numpy()
