"""Data-parallel sharding of the tokenize path across the GPUs of one node.

Each image's 32 ids depend only on that image (get_codebook_indices has no cross-batch op), so the batch is a
pure map: every rank holds a full weight replica (2.18 GB) and tokenizes a contiguous slice.  The reference's own
8-GPU tool does exactly this with no data collective at all, each rank writing its own shard
(MultiModalLLM/src/tools/extract_image_ids_to_torchdata_parallel.py:72-79,114-127).  The one exchange step this
framework adds is the gather of the int64 [B_local, 32] ids (64 KiB per rank at B_local = 256) — an RCCL
all-gather on the compute stream; latency-bound, one hop over the direct xGMI links.
One process per GPU; `backend="nccl"` is RCCL on ROCm, `gloo` on CPU for the tests.
"""
from typing import Optional

import torch


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous [begin, end) slice of ``n_items`` owned by ``rank`` (sizes differ by at most one)."""
    base, rem = divmod(n_items, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def gather_token_ids(ids: torch.Tensor, dist=None, group=None, always_collective: bool = False) -> torch.Tensor:
    """All-gather equal-sized [B_local, 32] id blocks into [world*B_local, 32] (rank order = shard order).
    ``always_collective`` issues the collective at world size 1 too (the RCCL readiness test: a single-GPU box still goes through
    ncclAllGather on the compute stream)."""
    if dist is None or not dist.is_initialized() or (dist.get_world_size(group) == 1 and not always_collective):
        return ids
    world = dist.get_world_size(group)
    out = torch.empty((world * ids.shape[0],) + tuple(ids.shape[1:]), dtype=ids.dtype, device=ids.device)
    dist.all_gather_into_tensor(out, ids.contiguous(), group=group)
    return out


def gather_ragged_token_ids(ids: torch.Tensor, n_total: int, dist, group=None) -> torch.Tensor:
    """All-gather for shard sizes from ``shard_range`` (may differ by one): pad to the max, gather, drop padding."""
    if dist is None or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return ids
    world = dist.get_world_size(group)
    per = (n_total + world - 1) // world
    pad = torch.zeros((per,) + tuple(ids.shape[1:]), dtype=ids.dtype, device=ids.device)
    pad[: ids.shape[0]] = ids
    full = gather_token_ids(pad, dist, group).view(world, per, *ids.shape[1:])
    parts = []
    for r in range(world):
        b, e = shard_range(n_total, r, world)
        parts.append(full[r, : e - b])
    return torch.cat(parts, 0)


def tokenize_data_parallel(encode_fn, images: torch.Tensor, dist=None, group=None) -> torch.Tensor:
    """Rank-local slice of a global batch -> ids of the WHOLE batch on every rank.
    ``encode_fn`` is the rank's engine (``TokenizerEngine.encode``); ``images`` is the global batch (only this
    rank's slice is touched)."""
    if dist is None or not dist.is_initialized():
        return encode_fn(images)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    b, e = shard_range(images.shape[0], rank, world)
    local = encode_fn(images[b:e])
    return gather_ragged_token_ids(local, images.shape[0], dist, group)
