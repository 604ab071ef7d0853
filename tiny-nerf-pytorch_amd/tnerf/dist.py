"""Multi-GPU plumbing: one process per GPU, rays sharded over ranks, ONE all-reduce(SUM) of the flat
fp32 gradient per step (RCCL over xGMI through torch.distributed's "nccl" backend; "gloo" on CPU for
tests).  The reference has no distributed code; SURVEY.md §8e defines the scheme:
every rank draws the same global (inds, t_rand), takes rows [lo, hi) and normalises its loss by the
GLOBAL element count, so the sum of the shard gradients is the full-batch gradient."""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_bounds(n: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous, balanced split of n rows: the first (n % world) ranks get one extra row."""
    base, extra = divmod(int(n), int(world_size))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_reduce_sum_(flat: torch.Tensor) -> torch.Tensor:
    """SUM over ranks, in place (a process group of one rank still goes through the collective: that is how the
    RCCL path is exercised on a single-GPU box)."""
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def broadcast_(flat: torch.Tensor, src: int = 0) -> torch.Tensor:
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(flat, src=src)
    return flat


def all_gather_rows(local: torch.Tensor, n_total: int) -> torch.Tensor:
    """Concatenate the ranks' row blocks (sizes from shard_bounds) into the full [n_total, ...] tensor."""
    rank, ws = world()
    if ws == 1:
        return local
    sizes = [shard_bounds(n_total, r, ws) for r in range(ws)]
    mx = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(ws)]
    dist.all_gather(parts, pad)
    return torch.cat([p[: hi - lo] for p, (lo, hi) in zip(parts, sizes)], dim=0)


def deal_round_robin(n: int, rank: int, world_size: int):
    """Whole work items (novel-view frames) dealt round-robin: rank r takes items r, r+world, ..."""
    return list(range(int(rank), int(n), int(world_size)))


def merge_round_robin(parts, n: int):
    """Inverse of deal_round_robin: parts[r] is rank r's list in its own order; returns the n items in global order."""
    world_size = len(parts)
    return [parts[k % world_size][k // world_size] for k in range(int(n))]
