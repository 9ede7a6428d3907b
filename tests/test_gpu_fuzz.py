"""The fuzz campaigns of rounds 4 - 5, inside the suite.

Round 5 ran them as scripts on the box (`scripts/gpu_fuzz_*.py`, ~35 000 scenarios, tallies under profiles/r05/); here one
slice of every campaign runs under `pytest -m gpu`, ~200 scenarios each on payload ranges the campaigns did not use, with
EVERY scenario held to the CPU oracle (the campaigns checked two of eight) besides the device-side comparison of the two
kernel families:

* the feed-forward range of the stage-parallel kernel (idle to saturated tandem servers, every latency law, spikes, outages);
* the round-4 / 5 range (servers in front of the LB, 4 - 5 server levels, 13 - 16 servers, tiers, general servers, tie storms);
* sweep columns over every accepted path, each point against the oracle on the payload a user of the reference would build;
* round 6: fractional RAM needs -- simpy's waiting `Container.put` and dead-locked RAM (next-event kernels' SimPy-order path).
"""

from __future__ import annotations

import importlib.util
import random
from pathlib import Path

import numpy as np
import pytest

from asyncflow_amd import _abi
from oracle import bulk

pytestmark = pytest.mark.gpu

SCRIPTS = Path(__file__).resolve().parent.parent / "scripts"


def _script(name: str):
    spec = importlib.util.spec_from_file_location(name, SCRIPTS / f"{name}.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_feed_forward_payloads_every_scenario_against_the_oracle():
    out = _script("gpu_fuzz_f3").run(25, 310_000, only="feed-forward", oracle_every=True)
    t = out["feed-forward payloads (tandem servers)"]
    assert out["different"] == 0 and t["scenarios"] + 8 * t["overflow_raised"] == 200 and t["oracle_checks"] == t["scenarios"]
    assert t["on_flow_kernel"] >= 0.9 * t["scenarios"]            # the family IS the stage-parallel kernel's range


def test_round_4_and_5_families_every_scenario_against_the_oracle():
    out = _script("gpu_fuzz_f3").run(4, 300_000, oracle_every=True)
    fams = [v for k, v in out.items() if k != "different"]
    assert out["different"] == 0 and len(fams) == 9             # round 6: + least connections over 9 .. 16 servers
    assert sum(t["scenarios"] + 8 * t["overflow_raised"] for t in fams) == 9 * 4 * 8
    assert sum(t["oracle_checks"] for t in fams) == sum(t["scenarios"] for t in fams) >= 200
    assert sum(t["on_flow_kernel"] for t in fams) >= 100          # (random topologies and tie storms mostly go to the next-event kernels)


def test_sweep_columns_every_point_against_the_oracle():
    t = _script("gpu_fuzz_sweeps").run(36, 320_000)
    assert t["different"] == 0, t["failures"]
    assert t["oracle_checks"] == t["scenarios"] >= 150 and t["columns"] >= t["payloads"]


def test_waiting_ram_puts_and_dead_locked_ram_on_the_device():
    """Fractional RAM needs (oracle/scenarios.py::fractional_ram_fuzz; the oracle is held to the live reference on this family
    in tests/test_reference_live.py): simpy refuses some `RAM.put`s by one rounding, responses wait for the next RAM get, a
    refused put facing a waiter that does not fit dead-locks the server's RAM.  Such plans run on the next-event kernels'
    SimPy-order path (af_core.hpp::m_srv_finish); every scenario against the oracle."""
    from asyncflow_amd.runner import SimulationRunner
    from oracle.scenarios import fractional_ram_fuzz

    scen = waits = dead = 0
    for case in range(28):
        payload = fractional_ram_fuzz(random.Random(424_200 + case), horizon=10)
        seeds = np.arange(8, dtype=np.uint64) + 900 + 1000 * case
        try:
            res = SimulationRunner(simulation_input=payload, seeds=seeds, on_negative_delay="flag").run()
        except OverflowError:          # a wait queue beyond the engine's maximum: reported, never silent
            continue
        # (a payload of the family whose endpoints ended up without a RAM step, or with a whole-MB one, may run on the
        # stage-parallel kernel; one with a decimal need never does: 1/256-MB needs only)
        want = bulk.simulate_many(payload, [int(s) for s in seeds])
        for i in range(8):
            w_counts, w_clock, w_samples, w_waits = want[i]
            got = res[i]
            assert got.counts[:5].astype(np.uint64).tolist() == w_counts[:5], (case, i, got.counts, w_counts)
            assert (int(got.counts[_abi.CNT_FLAGS]) & 0xFF) == (w_counts[_abi.CNT_FLAGS] & 0xFF), (case, i)
            assert bulk.digest_clock(got.rqs_clock) == w_clock, (case, i, "rqs_clock")
            assert bulk.digest_samples(got._samples) == w_samples, (case, i, "sampled series")  # noqa: SLF001
            scen += 1
            waits += w_waits > 0
            assert not (w_waits > 0 and int(res.engine_stats.flow_scenarios) != 0), (case, "a waiting put on the stage-parallel kernel")
            dead += (w_counts[_abi.CNT_FLAGS] & _abi.FLAG_RAM_STARVED) != 0
    assert scen >= 160 and waits >= 60 and dead >= 10


def test_a_saturated_server_with_more_than_16384_waiters_matches_the_oracle():
    """VERDICT r5 missing 3: the reference's simpy queues have no bound; the engine stopped at 16 384 waiters per server
    (OverflowError).  Round 6: a word per queue (af_core.hpp), up to 2^20 waiters, state in HBM, the runner grows the queue
    until the flag is clear.  300 arrivals/s into a 100-requests/s server for 120 s: 23 000 requests queue for the core."""
    import warnings

    from asyncflow_amd.runner import SimulationRunner
    from oracle.scenarios import _endpoint, _server, single_server

    payload = single_server(users=300, rpm=60, horizon=120, period=0.05)
    payload["topology_graph"]["nodes"]["servers"] = [_server("srv-1", 1, 2048, [_endpoint("/a", [("initial_parsing", 0.01), ("io_wait", 0.02)])])]
    seeds = np.arange(6, dtype=np.uint64) + 5
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)          # "engine capacity overflow ...; retrying with ..."
        res = SimulationRunner(simulation_input=payload, seeds=seeds).run()
    assert int(res.engine_stats.fifo_capacity) > 16_384 and (res.flags & _abi.FATAL_FLAGS).max() == 0
    want = bulk.simulate_many(payload, [int(s) for s in seeds])
    for i in range(len(seeds)):
        w_counts, w_clock, w_samples, _ = want[i]
        assert res[i].counts[:5].astype(np.uint64).tolist() == w_counts[:5]
        assert bulk.digest_clock(res[i].rqs_clock) == w_clock and bulk.digest_samples(res[i]._samples) == w_samples  # noqa: SLF001
        assert int(res[i]._samples[res.plan.n_edges].max()) > 16_384          # noqa: SLF001  (ready_queue_len of srv-1)
