"""Utterance sharding across the GPUs of a node.

The decoder loop shards by utterance: batch rows share only read-only weights, so every rank
(one process per GPU) decodes its own slice with its own handle and there is NO data-path
collective.  torch.distributed is used only for the barrier, the max-over-ranks time and the
host-side gather of results.

Host-side means host-side: every helper takes a `group=`; pass the gloo group that `host_group()`
creates next to an RCCL default group, and results travel over the hosts' sockets (RCCL then
carries the barrier only).  With the default group of an RCCL job `gather_object` would stage the
pickled results through GPU memory instead.
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


_HOST_GROUP = None


def host_group():
    """The process group host-side results travel on: the default group when it is gloo already, otherwise a gloo group
    over the same ranks, created once (collective: every rank must call it at the same point).  None without
    torch.distributed."""
    global _HOST_GROUP
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return None
    if _HOST_GROUP is None:
        _HOST_GROUP = dist.group.WORLD if dist.get_backend() == "gloo" else dist.new_group(backend="gloo")
    return _HOST_GROUP


def _reset_host_group():
    global _HOST_GROUP
    _HOST_GROUP = None


def _single(dist) -> bool:
    return not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1


def max_over_ranks(seconds: float, device: Optional[torch.device] = None, group=None) -> float:
    """Whole-job time = the slowest rank's time.  `group`: a gloo group (CPU tensor) or None = the default group, for
    which `device` must be the rank's GPU when that group is RCCL."""
    import torch.distributed as dist
    if _single(dist):
        return seconds
    on_host = group is not None and dist.get_backend(group) == "gloo"
    t = torch.tensor([seconds], dtype=torch.float64, device="cpu" if on_host else (device or "cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def gather_to_rank0(shard: torch.Tensor, dim: int = 0, group=None) -> Optional[torch.Tensor]:
    """Host-side gather of per-rank results (CPU tensors, possibly ragged along `dim`) on `group` (see host_group)."""
    import torch.distributed as dist
    shard = shard.detach().cpu()
    if _single(dist):
        return shard
    parts = [None] * dist.get_world_size() if dist.get_rank() == 0 else None
    dist.gather_object(shard, parts, dst=0, group=group)
    return torch.cat(parts, dim=dim) if dist.get_rank() == 0 else None


def gather_list_to_rank0(items: list, group=None) -> Optional[list]:
    """Host-side gather of per-rank python lists (e.g. one audio array per utterance), in rank order, on `group`."""
    import torch.distributed as dist
    if _single(dist):
        return list(items)
    parts = [None] * dist.get_world_size() if dist.get_rank() == 0 else None
    dist.gather_object(list(items), parts, dst=0, group=group)
    return [x for part in parts for x in part] if dist.get_rank() == 0 else None
