"""Multi-device plumbing on ONE MI355X: the RCCL gather of the C ABI at world size 1 and the in-process
`devices=` sharding with both shards on device 0 (the driver's 8-GPU run exercises N > 1; world-size-2
sharding / launcher logic is covered on CPU in tests/test_distributed_cpu.py)."""

from __future__ import annotations

import numpy as np
import pytest

from asyncflow_amd.workloads import lb_two_servers

pytestmark = pytest.mark.gpu


def test_engine_gather_through_the_c_abi_at_world_size_one():
    import torch

    from asyncflow_amd.distributed import EngineComm, gather_engine_summaries
    from asyncflow_amd.engine import Engine
    from asyncflow_amd.runner import SimulationRunner

    res = SimulationRunner(simulation_input=lb_two_servers(horizon=20), replicas=24).run()
    summ = res.summary(rps=True, hist_bins=64, hist_max=0.128, series=True)
    eng = Engine(res.plan, 0)
    comm = EngineComm(0, 1, 0)
    try:
        tensors = {k: summ[k] for k in ("stats", "rps", "hist", "series_mean", "series_max")}
        got = gather_engine_summaries(eng, comm, tensors, n_max=32)          # padded shard: 24 -> 32 rows
        assert eng.stats().gather_ms > 0.0
        for k, t in tensors.items():
            assert got[k].shape[0] == 32 and torch.equal(got[k][:24], t), k
            assert not got[k][24:].any()
    finally:
        comm.close()
        eng.close()


def test_devices_option_shards_in_process_and_keeps_the_sweep_order():
    from asyncflow_amd.runner import SimulationRunner

    payload = lb_two_servers(horizon=20)
    seeds = np.arange(30, dtype=np.uint64) + 400
    users = np.linspace(50, 600, 30)[np.random.default_rng(0).permutation(30)]
    sweep = {"rqs_input.avg_active_users.mean": users}
    one = SimulationRunner(simulation_input=payload, seeds=seeds, sweep=sweep).run()
    two = SimulationRunner(simulation_input=payload, seeds=seeds, sweep=sweep, devices=[0, 0]).run()
    assert len(two) == 30 and len(two.shards) == 2 and abs(len(two.shards[0]) - len(two.shards[1])) <= 1
    loads = [users[ix].sum() for ix in two.index]
    assert abs(loads[0] - loads[1]) / sum(loads) < 0.05            # dealt by expected load
    assert np.array_equal(two.counts[:, :6], one.counts[:, :6]) and np.array_equal(two.seeds, seeds)
    for i in (0, 7, 29):
        assert np.array_equal(two[i].rqs_clock, one[i].rqs_clock) and np.array_equal(two[i]._samples, one[i]._samples)  # noqa: SLF001
    a, b = one.summary(rps=True), two.summary(rps=True)
    assert np.array_equal(a["stats"].cpu().numpy(), b["stats"].cpu().numpy(), equal_nan=True)
    assert np.array_equal(a["rps"].cpu().numpy(), b["rps"].cpu().numpy())
    assert two.aggregate()["mean"]["p95"] == one.aggregate()["mean"]["p95"]
