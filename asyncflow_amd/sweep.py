"""Declarative sweep front-end: grids over YAML paths x seed replicas.

The reference runs one payload per ``SimulationRunner``; a parameter study there is a Python loop
that rebuilds the Pydantic payload for every point.  Here the payload is validated and lowered
ONCE and a sweep is a set of per-scenario columns (``resolve_sweep`` in runner.py lists the
accepted paths); this module only expands axes into those columns.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Mapping, Sequence

import numpy as np


@dataclass
class Sweep:
    """Columns of a flattened grid: ``columns[path][i]`` is the value of scenario ``i``."""

    columns: dict[str, np.ndarray]
    seeds: np.ndarray          #: uint64 Philox key per scenario
    point: np.ndarray          #: grid point index of scenario i (row-major over the axes)
    replica: np.ndarray        #: replica index of scenario i within its grid point
    shape: tuple[int, ...]     #: lengths of the axes, in the order given
    axes: dict[str, np.ndarray]

    def __len__(self) -> int:
        return int(self.seeds.size)

    def runner_kwargs(self) -> dict[str, Any]:
        """``SimulationRunner(simulation_input=payload, **sweep.runner_kwargs())``."""
        return {"seeds": self.seeds, "sweep": self.columns}

    def by_point(self, values: np.ndarray) -> np.ndarray:
        """Reshape per-scenario ``values`` [n, ...] to [*shape, replicas, ...] (sorted scenarios undone)."""
        values = np.asarray(values)
        reps = int(self.replica.max()) + 1 if len(self) else 0
        out = np.empty((int(np.prod(self.shape)), reps, *values.shape[1:]), dtype=values.dtype)
        out[self.point, self.replica] = values
        return out.reshape(*self.shape, reps, *values.shape[1:])


def expand_grid(axes: Mapping[str, Sequence[float]], *, replicas: int = 1, seed_base: int = 0xC0F30000,
                order_by_load: str | None = None) -> Sweep:
    """Cartesian product of ``axes`` (path -> values) x ``replicas`` seeds per point.

    Scenario ``i`` gets the Philox key ``seed_base + i`` (before any reordering), so a grid is
    reproducible from ``(axes, replicas, seed_base)``.  ``order_by_load`` names an axis (e.g.
    ``"rqs_input.avg_active_users.mean"``) by which scenarios are sorted, heaviest first: the engine
    packs consecutive scenarios into one wavefront, and lanes of similar length waste fewer rounds.
    """
    if replicas < 1:
        msg = "replicas must be >= 1"
        raise ValueError(msg)
    names = list(axes)
    vals = [np.asarray(axes[k], dtype=np.float64).reshape(-1) for k in names]
    if any(v.size == 0 for v in vals):
        msg = "every axis needs at least one value"
        raise ValueError(msg)
    shape = tuple(int(v.size) for v in vals)
    n_points = int(np.prod(shape)) if shape else 1
    mesh = np.meshgrid(*vals, indexing="ij") if vals else []
    point = np.repeat(np.arange(n_points, dtype=np.int64), replicas)
    replica = np.tile(np.arange(replicas, dtype=np.int64), n_points)
    columns = {k: np.repeat(m.reshape(-1), replicas) for k, m in zip(names, mesh)}
    seeds = (np.uint64(seed_base) + np.arange(n_points * replicas, dtype=np.uint64)).astype(np.uint64)
    if order_by_load is not None:
        if order_by_load not in columns:
            msg = f"order_by_load: {order_by_load!r} is not an axis"
            raise ValueError(msg)
        order = np.argsort(-columns[order_by_load], kind="stable")
        columns = {k: np.ascontiguousarray(v[order]) for k, v in columns.items()}
        seeds, point, replica = seeds[order], point[order], replica[order]
    return Sweep(columns=columns, seeds=np.ascontiguousarray(seeds), point=point, replica=replica, shape=shape,
                 axes=dict(zip(names, vals)))
