"""Batch sharding for multi-GPU encode/decode: vectors are independent, so each rank encodes a
contiguous shard with a replicated quantizer state and there is no data-path collective
(SURVEY.md 8e).  `encode_sharded` is the one-process-per-GPU entry point."""
from typing import Optional, Tuple

import torch


def shard_bounds(total: int, world: int, rank: int) -> Tuple[int, int]:
    """[begin, end) of rank's contiguous shard; sizes differ by at most one."""
    base, rem = divmod(total, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def encode_sharded(quantizer, x: torch.Tensor, refine_indexes_iters: int = 5, as_bytes: bool = True,
                   group=None, gather: bool = False) -> torch.Tensor:
    """Every rank holds (or can index) the whole batch x (*, dim) on its own device and encodes
    only its shard.  Returns the local shard's codes, or -- with gather=True -- the codes of the
    whole batch on every rank (the only collective, and only on request: all_gather of N bytes
    per vector)."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    flat = x.reshape(-1, quantizer.dim)
    lo, hi = shard_bounds(flat.shape[0], world, rank)
    codes = quantizer.encode(flat[lo:hi], refine_indexes_iters, as_bytes)
    if not gather or world == 1:
        return codes
    # all_gather needs equal sizes: pad every shard to the largest (sizes differ by at most one row)
    sizes = [b - a for a, b in (shard_bounds(flat.shape[0], world, r) for r in range(world))]
    width = max(sizes)
    padded = torch.zeros((width, codes.shape[-1]), dtype=codes.dtype, device=codes.device)
    padded[:codes.shape[0]] = codes
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([p[:n] for p, n in zip(parts, sizes)], dim=0)
