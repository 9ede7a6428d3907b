"""The C oracle over MANY scenarios at once, on every host core -- ORACLE / TEST INFRASTRUCTURE ONLY.

Only tests/ and bench.py's checker legs may import this module; ``asyncflow_amd`` never does.

A GPU parity test that holds a benched batch to the oracle one scenario after the other gets through ~25 LB-2 scenarios per
second.  This module runs `oracle_lib.simulate` in a pool of SPAWNED processes (the caller has a HIP context: forking it is
not safe) and returns, per scenario, the counts and a 128-bit digest of the `rqs_clock` rows and of the sampled series -- a
few dozen bytes instead of 1.8 MB through the pipe; the caller digests the device's own outputs the same way
(`digest_clock` / `digest_samples`).  A differing digest is a differing scenario: the caller re-runs that one through
`oracle_lib.simulate` for the details.
"""

from __future__ import annotations

import json
import os
from concurrent.futures import ProcessPoolExecutor
from typing import Iterable, Sequence

import numpy as np
import xxhash


def digest_clock(clock: np.ndarray) -> bytes:
    """[completed][2] float64 (start, finish) rows in completion order -> 16 bytes."""
    return xxhash.xxh3_128_digest(np.ascontiguousarray(clock, dtype=np.float64).view(np.uint8))


def digest_samples(samples_by_series: np.ndarray) -> bytes:
    """[n_series][ticks] 4-byte words (the oracle's layout; ram rows are float32 bits) -> 16 bytes."""
    return xxhash.xxh3_128_digest(np.ascontiguousarray(samples_by_series).view(np.uint8))


_worker_payload: dict | None = None


def _init(payload_json: str) -> None:
    global _worker_payload  # noqa: PLW0603
    _worker_payload = json.loads(payload_json)
    from oracle import oracle_lib as ol

    ol.lib()


def _job(job: tuple) -> list[tuple]:
    """One chunk of scenarios: (seed, overrides [(name, index, value)], clock capacity) each."""
    from asyncflow_amd.plan import lower
    from oracle import oracle_lib as ol

    out = []
    for seed, overrides, clock_cap in job:
        plan = lower(_worker_payload)
        if overrides:
            ol.apply_overrides(plan, {(name, int(idx)): float(v) for name, idx, v in overrides})
        r = ol.simulate(plan, int(seed), clock_capacity=clock_cap)
        out.append((r.counts.tolist(), digest_clock(r.clock), digest_samples(r.samples), r.put_waits))
    return out


def usable_cores() -> int:
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:   # cgroup v2 quota ("max" or "<quota> <period>")
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def simulate_many(payload: dict, seeds: Sequence[int], overrides: Iterable[list[tuple[str, int, float]]] | None = None,
                  clock_capacity: int | None = None, procs: int | None = None) -> list[tuple[list[int], bytes, bytes, int]]:
    """The oracle on every (seed, overrides) of one payload: [(counts[8], clock digest, samples digest, waiting RAM puts)],
    in the order given.  ``overrides``: per scenario the sweep columns' values as `oracle_lib.apply_overrides` names them."""
    seeds = [int(s) for s in seeds]
    over = list(overrides) if overrides is not None else [[] for _ in seeds]
    assert len(over) == len(seeds)
    procs = procs or usable_cores()
    jobs_flat = [(s, o, clock_capacity) for s, o in zip(seeds, over)]
    if procs <= 1 or len(seeds) < 4:
        _init(json.dumps(payload))
        return _job(tuple(jobs_flat))
    per = max(1, min(16, len(seeds) // (4 * procs)))          # (chunks small enough to balance unlike scenarios)
    chunks = [tuple(jobs_flat[i:i + per]) for i in range(0, len(jobs_flat), per)]
    import multiprocessing as mp

    with ProcessPoolExecutor(max_workers=min(procs, len(chunks)), mp_context=mp.get_context("spawn"),
                             initializer=_init, initargs=(json.dumps(payload),)) as pool:
        parts = list(pool.map(_job, chunks))
    return [row for part in parts for row in part]
