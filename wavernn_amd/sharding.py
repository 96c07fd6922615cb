"""Utterance sharding across the GPUs of a node.

The decoder loop shards by utterance: batch rows share only read-only weights, so every rank
(one process per GPU) decodes its own slice with its own handle and there is NO data-path
collective.  torch.distributed is used only for the barrier, the max-over-ranks time and the
host-side gather of results.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch


def partition(n_items: int, world_size: int) -> List[Tuple[int, int]]:
    """Contiguous, balanced [lo, hi) ranges; the first n_items % world_size ranks get one extra."""
    if world_size < 1 or n_items < 0:
        raise ValueError("world_size must be >= 1 and n_items >= 0")
    q, r = divmod(n_items, world_size)
    out, lo = [], 0
    for k in range(world_size):
        hi = lo + q + (1 if k < r else 0)
        out.append((lo, hi))
        lo = hi
    return out


def my_slice(n_items: int, rank: int, world_size: int) -> Tuple[int, int]:
    return partition(n_items, world_size)[rank]


def max_over_ranks(seconds: float, device: Optional[torch.device] = None) -> float:
    """Whole-job time = the slowest rank's time."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_to_rank0(shard: torch.Tensor, dim: int = 0) -> Optional[torch.Tensor]:
    """Host-side gather of per-rank results (CPU tensors, possibly ragged along `dim`)."""
    import torch.distributed as dist
    shard = shard.detach().cpu()
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return shard
    parts = [None] * dist.get_world_size() if dist.get_rank() == 0 else None
    dist.gather_object(shard, parts, dst=0)
    return torch.cat(parts, dim=dim) if dist.get_rank() == 0 else None


def gather_list_to_rank0(items: list) -> Optional[list]:
    """Host-side gather of per-rank python lists (e.g. one audio array per utterance), in rank order."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return list(items)
    parts = [None] * dist.get_world_size() if dist.get_rank() == 0 else None
    dist.gather_object(list(items), parts, dst=0)
    return [x for part in parts for x in part] if dist.get_rank() == 0 else None
