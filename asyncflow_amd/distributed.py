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


class EngineComm:
    """A RCCL communicator made through the engine's C ABI (``af_comm_*``): what a non-Python host would use.

    The 128-byte id is created on rank 0 and shared out of band: through the initialised
    ``torch.distributed`` process group (any backend) when there is one, else through ``share(id_bytes | None)
    -> id_bytes`` supplied by the caller.  The engine resolves RCCL at run time; under PyTorch it is pointed
    at torch's own copy (torch/lib/librccl.so) so that one RCCL lives in the process.
    """

    def __init__(self, rank: int, world_size: int, device: int, share: Any = None) -> None:
        import ctypes as C
        from pathlib import Path

        from . import _abi
        from .engine import EngineError, load_library

        self._lib = lib = load_library()
        self.rank, self.world_size = int(rank), int(world_size)
        try:
            import torch

            cand = Path(torch.__file__).resolve().parent / "lib" / "librccl.so"
            if cand.exists():
                lib.af_comm_load(str(cand).encode())
        except ImportError:  # pragma: no cover - torch is plumbing
            pass
        buf = C.create_string_buffer(_abi.COMM_ID_BYTES)
        if self.rank == 0 and lib.af_comm_unique_id(buf) != _abi.AF_OK:
            raise EngineError(f"af_comm_unique_id: {(lib.af_last_error() or b'').decode()}")
        ident = bytes(buf.raw) if self.rank == 0 else None
        if share is not None:
            ident = share(ident)
        elif self.world_size > 1:
            import torch.distributed as dist

            box = [ident]
            dist.broadcast_object_list(box, src=0)
            ident = box[0]
        handle = C.c_void_p()
        rc = lib.af_comm_init_rank(C.create_string_buffer(ident, _abi.COMM_ID_BYTES), self.world_size, self.rank, int(device),
                                   C.byref(handle))
        if rc != _abi.AF_OK:
            raise EngineError(f"af_comm_init_rank: {(lib.af_last_error() or b'').decode()}")
        self.handle = handle

    def count(self) -> tuple[int, int]:
        """(ranks, this rank) as RCCL reports them for the communicator (``ncclCommCount`` / ``ncclCommUserRank``)."""
        import ctypes as C

        from . import _abi
        from .engine import EngineError

        n, r = C.c_int(0), C.c_int(0)
        if self._lib.af_comm_count(self.handle, C.byref(n), C.byref(r)) != _abi.AF_OK:
            raise EngineError(f"af_comm_count: {(self._lib.af_last_error() or b'').decode()}")
        return int(n.value), int(r.value)

    def close(self) -> None:
        if getattr(self, "handle", None):
            self._lib.af_comm_destroy(self.handle)
            self.handle = None

    def __del__(self) -> None:  # pragma: no cover - best effort
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


def gather_engine_summaries(engine: Any, comm: EngineComm, tensors: dict[str, Any], n_max: int) -> dict[str, Any]:
    """All-gather the summary tensors of this rank through ``af_engine_gather`` (RCCL via the C ABI).

    ``tensors``: ``stats`` f64 [n, 8], ``rps`` f32 [n, T], ``hist`` i32 [n, bins], optional ``series_mean`` /
    ``series_max`` (None = skipped); shards shorter than ``n_max`` (the longest shard of the job) are
    zero-padded.  Returns tensors of ``world * n_max`` rows, rank order."""
    import torch

    local, out, padded, result = {}, {}, {}, {}
    for k, t in tensors.items():
        if t is None:
            continue
        if t.shape[0] < n_max:
            t = torch.cat([t, torch.zeros((n_max - t.shape[0], *t.shape[1:]), dtype=t.dtype, device=t.device)], dim=0)
        padded[k] = t.contiguous()
        result[k] = torch.empty((comm.world_size * n_max, *t.shape[1:]), dtype=t.dtype, device=t.device)
        local[k], out[k] = padded[k].data_ptr(), result[k].data_ptr()
    torch.cuda.synchronize()
    engine.gather(comm.handle, comm.world_size, n_max, local, out,
                  rps_buckets=int(padded["rps"].shape[1]) if "rps" in padded else 0,
                  hist_bins=int(padded["hist"].shape[1]) if "hist" in padded else 0)
    return result
