"""Sharding of independent chunks / channels over ranks.

Chunks are self-contained blosc2 buffers, so the codec path needs no data-path collective: every rank
compresses / decompresses its own items with its own engine.  The only exchanges are tiny: the
max-over-ranks of the elapsed time, and the per-item compressed sizes so that every rank (or rank 0)
knows the global layout.  Works with any torch.distributed backend ("nccl" = RCCL on the GPU box, "gloo"
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
                  device="cpu"):
    """The one real exchange step of SURVEY.md section 8e: owner ranks send their finished chunks to `dst`.

    local_buf   uint8 torch tensor (or numpy array) holding this rank's chunks, chunk k of `local_idx` at
                local_off[k] with length sizes_all[local_idx[k]]
    sizes_all   result of gather_sizes() (every rank knows every chunk's compressed size)
    Returns on `dst` a list of n_items uint8 numpy arrays in global chunk order, elsewhere None.
    Payloads travel packed back to back (only compressed bytes, C/world per rank), padded to the largest
    per-rank total because gather needs equal shapes; with backend "nccl" this is RCCL send/recv over xGMI.
    It is a property of a caller that wants all chunks on one device -- bench.py never calls it.
    """
    import torch
    sizes_all = np.asarray(sizes_all, dtype=np.int64)
    owners = [partition(n_items, world, r, items_per_group) for r in range(world)]
    totals = [int(sizes_all[o].sum()) for o in owners]
    pad = max(totals + [1])
    if not torch.is_tensor(local_buf):
        local_buf = torch.as_tensor(np.ascontiguousarray(local_buf).view(np.uint8))
    packed = torch.zeros(pad, dtype=torch.uint8, device=device)
    at = 0
    for k, g in enumerate(np.asarray(local_idx, dtype=np.int64)):
        n = int(sizes_all[g])
        packed[at:at + n] = local_buf[int(local_off[k]):int(local_off[k]) + n].to(device)
        at += n
    if dist is None or not dist.is_initialized() or world == 1:
        parts = [packed]
    else:
        parts = [torch.empty(pad, dtype=torch.uint8, device=device) for _ in range(world)] if rank == dst else None
        dist.gather(packed, gather_list=parts, dst=dst)
    if rank != dst:
        return None
    out = [None] * n_items
    for r in range(world):
        host = parts[r].cpu().numpy()
        at = 0
        for g in owners[r]:
            n = int(sizes_all[g])
            out[int(g)] = host[at:at + n].copy()
            at += n
    return out
