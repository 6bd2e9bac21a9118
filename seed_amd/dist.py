"""Data-parallel sharding of the tokenize path across the GPUs of one node.

Each image's 32 ids depend only on that image (get_codebook_indices has no cross-batch op), so the batch is a
pure map: every rank holds a full weight replica (2.18 GB) and tokenizes a contiguous slice.  The reference's own
8-GPU tool does exactly this with no data collective at all, each rank writing its own shard
(MultiModalLLM/src/tools/extract_image_ids_to_torchdata_parallel.py:72-79,114-127).  The one exchange step this
framework adds is the gather of the [B_local, 32] ids — sent as int16 (16 KiB per rank at B_local = 256, ids < 8192) and widened to
int64 on arrival — an RCCL all-gather on the compute stream; latency-bound, one hop over the direct xGMI links.
One process per GPU; `backend="nccl"` is RCCL on ROCm, `gloo` on CPU for the tests.
"""
from typing import Optional

import torch


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous [begin, end) slice of ``n_items`` owned by ``rank`` (sizes differ by at most one)."""
    base, rem = divmod(n_items, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


WIRE_DTYPE = torch.int16            # token ids are < 8192 (n_embed): a quarter of the int64 bytes on the links (16 KiB per rank at 256 images)


def gather_token_ids(ids: torch.Tensor, dist=None, group=None, always_collective: bool = False, wire_int16: bool = True) -> torch.Tensor:
    """All-gather equal-sized [B_local, 32] id blocks into [world*B_local, 32] (rank order = shard order).
    The ids travel as int16 and are widened to the caller's dtype on arrival (``wire_int16``; values must lie in [-32768, 32767] - VQ codes
    are in [0, 8192) - anything else is refused rather than truncated); the collective is issued on the CURRENT (compute) stream, so it is
    ordered behind the tokenize kernels without a host sync.  ``always_collective`` issues the collective at world size 1 too (the RCCL
    readiness test: a single-GPU box still goes through ncclAllGather on the compute stream)."""
    if dist is None or not dist.is_initialized() or (dist.get_world_size(group) == 1 and not always_collective):
        return ids
    world = dist.get_world_size(group)
    narrow = wire_int16 and ids.dtype in (torch.int64, torch.int32)
    if narrow and ids.numel():
        if not ids.is_cuda:
            if int(ids.min()) < -32768 or int(ids.max()) > 32767:
                raise ValueError("gather_token_ids: ids outside the int16 wire range; pass wire_int16=False")
        else:
            # device tensors: no host sync on the path - an asynchronous device-side assertion (a kernel that traps the stream if it fails),
            # so an id >= 32768 (e.g. LLM-vocabulary ids 32000 + code) can never wrap silently
            wide = ids.to(torch.int32)
            torch._assert_async(((wide >= -32768) & (wide <= 32767)).all(),
                                "gather_token_ids: ids outside the int16 wire range; pass wire_int16=False")
    if not narrow:
        send = ids.contiguous()
        out = torch.empty((world * ids.shape[0],) + tuple(ids.shape[1:]), dtype=send.dtype, device=ids.device)
        dist.all_gather_into_tensor(out, send, group=group)
        return out
    # int16 is not a collective data type of either backend (ncclDataType_t has no 16-bit integer, gloo refuses it): an all-gather moves
    # bytes, so the int16 block travels as its uint8 view and is re-viewed on arrival
    send = ids.contiguous().to(WIRE_DTYPE).view(torch.uint8).reshape(-1)
    out = torch.empty(world * send.numel(), dtype=torch.uint8, device=ids.device)
    dist.all_gather_into_tensor(out, send, group=group)
    return out.view(WIRE_DTYPE).view((world * ids.shape[0],) + tuple(ids.shape[1:])).to(ids.dtype)


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
