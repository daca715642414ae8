"""Multi-GPU layout: one process per GPU, contiguous shards of the global batch, no data-path
collective.  The only exchange is the episodic-return all-gather once per rollout (RCCL over xGMI
through ``torch.distributed``; backend "nccl" is RCCL on ROCm, "gloo" in the CPU tests).

The reference has no distributed code at all; this follows BASELINE.json's
north_star: shard the batch over the GPUs of one node, all-gather only the episodic returns.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.distributed as dist


@dataclass(frozen=True)
class Shard:
    rank: int
    world_size: int
    n_global: int
    offset: int   # global index of this rank's first board
    n_local: int

    @property
    def stop(self) -> int:
        return self.offset + self.n_local


def shard_range(n_global: int, rank: int, world_size: int) -> Shard:
    """Contiguous split; the first ``n_global % world_size`` ranks get one extra board."""
    if not 0 <= rank < world_size:
        raise ValueError("rank out of range")
    base, extra = divmod(n_global, world_size)
    n_local = base + (1 if rank < extra else 0)
    offset = rank * base + min(rank, extra)
    return Shard(rank, world_size, n_global, offset, n_local)


def weak_shard(n_per_gpu: int, rank: int, world_size: int) -> Shard:
    """Weak scaling: every rank owns ``n_per_gpu`` boards of a ``world_size * n_per_gpu`` batch."""
    return Shard(rank, world_size, n_per_gpu * world_size, rank * n_per_gpu, n_per_gpu)


def allgather_returns(local_returns: torch.Tensor, shard: Shard) -> torch.Tensor:
    """All-gather per-board episodic returns into the global order (``[n_global]`` on every rank).

    ``local_returns``: ``[n_local]`` tensor on this rank's device (e.g. ``Batched2048.last_scores()``).
    Equal shards use one ``all_gather_into_tensor`` (one RCCL all-gather over xGMI); ragged shards
    pad to the largest shard first.
    """
    if shard.world_size == 1 or not dist.is_initialized():
        return local_returns.clone()
    base, extra = divmod(shard.n_global, shard.world_size)
    n_max = base + (1 if extra else 0)
    send = local_returns
    if send.numel() != n_max:
        send = torch.zeros(n_max, dtype=local_returns.dtype, device=local_returns.device)
        send[: local_returns.numel()] = local_returns
    out = torch.empty(n_max * shard.world_size, dtype=send.dtype, device=send.device)
    dist.all_gather_into_tensor(out, send.contiguous())
    if extra == 0:
        return out
    parts = [out[r * n_max: r * n_max + shard_range(shard.n_global, r, shard.world_size).n_local]
             for r in range(shard.world_size)]
    return torch.cat(parts)


def allreduce_summary(episodes: int, score_sum: int, max_score: int, device) -> dict:
    """Global (episodes, score_sum, max_score) from per-rank episode statistics."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return dict(episodes=episodes, score_sum=score_sum, max_score=max_score)
    sums = torch.tensor([episodes, score_sum], dtype=torch.int64, device=device)
    mx = torch.tensor([max_score], dtype=torch.int64, device=device)
    dist.all_reduce(sums, op=dist.ReduceOp.SUM)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    return dict(episodes=int(sums[0]), score_sum=int(sums[1]), max_score=int(mx[0]))
