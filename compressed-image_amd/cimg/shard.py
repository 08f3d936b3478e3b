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
