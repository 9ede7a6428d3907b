"""N > 1 path on CPU: 2 processes, gloo backend, the same sharding + gather code bench.py uses."""

from __future__ import annotations

import os
import socket

import numpy as np
import pytest

from asyncflow_amd.distributed import interleave_by_load, shard_bounds, shard_seeds


def test_shards_partition_the_seed_range():
    seeds = 0x5EED0000 + np.arange(1003, dtype=np.uint64)
    for world in (1, 2, 3, 8):
        parts = [shard_seeds(seeds, r, world) for r in range(world)]
        assert np.array_equal(np.concatenate(parts), seeds)
        sizes = [len(p) for p in parts]
        assert max(sizes) - min(sizes) <= 1
        assert [shard_bounds(1003, r, world) for r in range(world)][0][0] == 0


def test_interleave_balances_expected_load():
    users = np.repeat(np.arange(1, 101) * 10.0, 100)       # config 3: events ~ users
    parts = interleave_by_load(users, 8)
    loads = [users[p].sum() for p in parts]
    assert sorted(np.concatenate(parts).tolist()) == list(range(10_000))
    assert (max(loads) - min(loads)) / np.mean(loads) < 0.01


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, n_total: int) -> None:
    import torch
    import torch.distributed as dist

    from asyncflow_amd import _abi
    from asyncflow_amd.distributed import gather_summaries, shard_seeds
    from asyncflow_amd.plan import lower
    from oracle import oracle_lib as ol
    from oracle.scenarios import lb_two_servers

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        plan = lower(lb_two_servers(horizon=5))
        seeds = 0x5EED0000 + np.arange(n_total, dtype=np.uint64)
        mine = shard_seeds(seeds, rank, world)

        def summary(seed: int) -> list[float]:
            r = ol.simulate(plan, int(seed))      # (the checker stands in for the engine: the test is about sharding + gather,
            counts, clock = r.counts, r.clock     #  and must not depend on whether the kernel headers compile)
            lat = clock[:, 1] - clock[:, 0]
            return [float(seed), float(counts[_abi.CNT_GENERATED]), float(counts[_abi.CNT_COMPLETED]),
                    float(np.percentile(lat, 95)) if len(lat) else float("nan")]

        local = torch.tensor([summary(s) for s in mine], dtype=torch.float64).reshape(len(mine), 4)
        full = gather_summaries(local)                            # the single collective
        assert full.shape == (n_total, 4)
        assert np.array_equal(full[:, 0].numpy().astype(np.uint64), seeds)      # rank order == seed order
        want = torch.tensor([summary(s) for s in seeds], dtype=torch.float64)
        assert torch.equal(torch.nan_to_num(full), torch.nan_to_num(want))
        total = torch.tensor([local[:, 2].sum()], dtype=torch.float64)
        dist.all_reduce(total)
        assert float(total) == float(want[:, 2].sum())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [6, 7])   # equal and ragged shards
def test_two_rank_gloo_gather_matches_serial(n_total):
    import torch.multiprocessing as mp

    from oracle import oracle_lib as ol

    ol.build()      # before the ranks start, so that they do not race to compile it
    mp.spawn(_worker, args=(2, _free_port(), n_total), nprocs=2, join=True)


@pytest.mark.parametrize("config, scenarios", [(4, 1800), (5, 1001)])
def test_bench_launcher_shards_the_stated_total_over_two_ranks(config, scenarios):
    """`python bench.py --gpus 2` spawns its own ranks (torch.distributed.run, 127.0.0.1), shards the
    STATED total (not a per-rank copy), gathers once and prints ONE line with n_gpus = 2.  --selftest-cpu
    swaps the engine for fabricated summary rows and RCCL for gloo; everything else is the bench's own path."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--selftest-cpu", "--config", str(config),
                          "--scenarios", str(scenarios)], capture_output=True, text=True, timeout=300, env=env, check=False)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["selftest"] is True
    assert line["scenarios_total"] == line["unique_seeds"] == sum(line["scenarios_per_rank"])
    assert abs(line["scenarios_per_rank"][0] - line["scenarios_per_rank"][1]) <= 1
    if config == 4:      # the grid is dealt by expected load: both ranks carry the same users x T mass
        a, b = line["load_per_rank"]
        assert abs(a - b) / (a + b) < 0.01
        assert line["scenarios_total"] == 42 * 42          # side^2 grid points x 1 seed
    else:
        assert line["scenarios_total"] == scenarios
