"""Multi-GPU plumbing: scenario sharding + the single end-of-run collective.

The path shards embarrassingly (one ``env`` per scenario in the reference,
docs/api/high-level/runner.md:203-207): every rank simulates its own seed range
and NO collective runs during simulation.  The only exchange is one all-gather
of fixed-size per-scenario summaries (RCCL over xGMI with backend "nccl"; "gloo"
on CPU in the tests).
"""

from __future__ import annotations

from typing import Any

import numpy as np


def shard_bounds(n_total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous, balanced scenario range of ``rank`` (first ``n_total % world`` ranks get one more)."""
    base, extra = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_seeds(seeds: np.ndarray, rank: int, world: int) -> np.ndarray:
    lo, hi = shard_bounds(len(seeds), rank, world)
    return np.ascontiguousarray(seeds[lo:hi])


def interleave_by_load(expected_events: np.ndarray, world: int) -> list[np.ndarray]:
    """Grid sweeps: deal scenarios to ranks in order of expected event count so that every
    rank gets the same mix of light and heavy scenarios (events ~ users x T)."""
    order = np.argsort(-np.asarray(expected_events, dtype=np.float64), kind="stable")
    return [np.sort(order[r::world]) for r in range(world)]


def gather_summaries(local: Any, n_per_rank: list[int] | None = None, group: Any = None) -> Any:
    """All-gather per-scenario summary rows ``[n_local, k]`` into ``[n_total, k]`` (rank order).

    Equal shard sizes use one ``all_gather_into_tensor``; ragged shards are padded to the
    largest shard and trimmed afterwards.  Without an initialised process group the input
    is returned unchanged.
    """
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return local
    world = dist.get_world_size(group)
    n_local = int(local.shape[0])
    if n_per_rank is None:
        sizes = torch.zeros(world, dtype=torch.int64, device=local.device)
        sizes[dist.get_rank(group)] = n_local
        dist.all_reduce(sizes, group=group)
        n_per_rank = [int(x) for x in sizes.tolist()]
    n_max = max(n_per_rank)
    if n_local < n_max:
        pad = torch.zeros((n_max - n_local, *local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], dim=0)
    local = local.contiguous()
    out = torch.empty((world * n_max, *local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local, group=group)
    if all(n == n_max for n in n_per_rank):
        return out
    parts = [out[r * n_max: r * n_max + n_per_rank[r]] for r in range(world)]
    return torch.cat(parts, dim=0)
