"""CPU differential tests of the ENGINE CORE (asyncflow_amd/csrc/af_core.hpp).

tests/hostcheck/ compiles the very state machine the HIP kernel runs for one
lane with g++ (test-only, never shipped).  It must reproduce the SimPy-faithful
oracle (oracle/des_oracle.c) -- and thereby the reference itself (tests/golden/,
tests/test_reference_live.py) -- bit for bit: counts, every (start, finish)
pair in completion order, every sampled value.  That includes instants shared
by several timed events, which the core runs through its SimPy-order path
(`micro_mode`); AF_FLAG_TIME_TIE must stay clear.
"""

from __future__ import annotations

import json
import random

import numpy as np
import pytest

from asyncflow_amd import _abi
from asyncflow_amd.plan import lower
from oracle import oracle_lib as ol
from oracle.scenarios import (lb_two_servers, overload, random_payload, server_chain, stress_mixed, tie_storm,
                              wide_fanout)
from tests.conftest import GOLDEN_DIR, golden_names
from tests.hostcheck import build as hc


def _assert_same(plan, seed, **kw):
    a = ol.simulate(plan, seed)
    counts, clock, samples = hc.simulate(plan, seed, **kw)
    assert np.array_equal(a.counts[:5].astype(np.uint32), counts[:5])
    assert int(counts[_abi.CNT_MARKS]) == int(a.counts[_abi.CNT_MARKS])
    assert np.array_equal(a.clock.view(np.uint64), clock.view(np.uint64))
    assert np.array_equal(a.samples, samples)
    assert (int(counts[_abi.CNT_FLAGS]) & (_abi.FATAL_FLAGS | _abi.FLAG_TIME_TIE)) == 0
    return a, counts


@pytest.mark.parametrize("name", golden_names())
def test_core_reproduces_the_reference_fixtures(name):
    fx = np.load(GOLDEN_DIR / f"{name}.npz", allow_pickle=False)
    plan = lower(json.loads(str(fx["payload_json"])))
    a, counts = _assert_same(plan, int(fx["seed"]))
    if not int(counts[_abi.CNT_FLAGS]) & _abi.FLAG_RAM_STARVED:     # (requests that wait for good are not kept by the engine)
        assert int(counts[_abi.CNT_MAX_LIVE]) == int(a.counts[_abi.CNT_MAX_LIVE])
    _, clock, samples = hc.simulate(plan, int(fx["seed"]))
    assert np.array_equal(clock, fx["clock"]) and np.array_equal(samples, fx["samples"])   # the reference's own output


@pytest.mark.parametrize("case", range(40))
def test_core_matches_oracle_on_fuzzed_payloads(case):
    rng = random.Random(4000 + case)
    _assert_same(lower(random_payload(rng, horizon=8)), 77 + case)


def test_shared_timestamps_follow_simpy_order():
    """Grant bursts on multi-core servers, integer (Poisson) edge latencies incl. zero, ticks on
    timeline marks: tens of thousands of timed events share their instant with another one."""
    ties = 0
    for case in range(60):
        rng = random.Random(777000 + case)
        a, _ = _assert_same(lower(tie_storm(rng, horizon=12)), 5 + case)
        ties += a.ties
    assert ties > 5000
    a, _ = _assert_same(lower(overload(horizon=30)), 5)
    assert a.ties > 0
    a, _ = _assert_same(lower(stress_mixed(40)), 3)
    assert a.ties > 100


def test_lean_first_pass_then_simpy_order_rerun_gives_the_same_results():
    """af_engine_run simulates with the lean kernel variant first and repeats, with the variant that
    has the SimPy-order path, the scenarios that met a shared instant."""
    L = hc.lib()
    L.hc_set_two_pass(1)
    try:
        before = L.hc_reruns()
        for case in range(40):
            _assert_same(lower(tie_storm(random.Random(777000 + case), horizon=12)), 5 + case)
        for case in range(20):
            _assert_same(lower(random_payload(random.Random(4000 + case), horizon=8)), 77 + case)
        _assert_same(lower(lb_two_servers(horizon=20)), 1)
        assert L.hc_reruns() - before >= 30          # the storms all need the second pass ...
        before = L.hc_reruns()
        _assert_same(lower(lb_two_servers(horizon=20)), 2)
        assert L.hc_reruns() == before               # ... the BASELINE topology does not
    finally:
        L.hc_set_two_pass(0)


@pytest.mark.parametrize(("n_srv", "algo"), [(9, "round_robin"), (20, "least_connection"), (40, "round_robin")])
def test_more_than_eight_servers_behind_the_load_balancer(n_srv, algo):
    """The rotation list no longer fits one register: it lives in state memory (outages edit it)."""
    _assert_same(lower(wide_fanout(n_srv, algo)), 1)


@pytest.mark.parametrize(("dist", "mean"), [("exponential", 0.003), ("poisson", 0.7), ("poisson", 0.2), ("normal", 0.001)])
def test_server_to_server_chain_with_zero_delay_hops(dist, mean):
    for seed in (1, 2, 3):
        a, _ = _assert_same(lower(server_chain(dist, mean)), seed)
        assert a.ties > 100


def test_kernel_side_summary_counts_exactly():
    """af_outputs_t.online_hist / online_rps: the histogram and the 1-s windows the kernel keeps itself
    equal the analyzer oracle's on the scenario's own rqs_clock (also through the two-pass flow)."""
    import ctypes as C

    from oracle import analyzer_oracle as ao

    L = hc.lib()
    u32p = C.POINTER(C.c_uint32)
    for two_pass, payload, seed in ((0, lb_two_servers(horizon=20), 3), (1, tie_storm(random.Random(777002), horizon=12), 7),
                                    (0, stress_mixed(30), 1)):
        plan = lower(payload)
        bins, hist_max, buckets = 128, 0.5 if not two_pass else 8.0, int(plan.total_time)
        hist = np.full(bins, 7, dtype=np.uint32)
        rps = np.full(buckets, 7, dtype=np.uint32)
        L.hc_set_two_pass(two_pass)
        L.hc_set_online(hist.ctypes.data_as(u32p), bins, hist_max, rps.ctypes.data_as(u32p), buckets)
        try:
            _, clock, _ = hc.simulate(plan, seed)
        finally:
            L.hc_set_online(None, 0, 1.0, None, 0)
            L.hc_set_two_pass(0)
        assert len(clock) >= 3
        assert np.array_equal(hist, ao.latency_histogram(clock, bins, hist_max))
        assert np.array_equal(rps.astype(np.float64), ao.throughput_series(clock, plan.total_time)[1])


def test_overrides_are_applied_per_scenario():
    payload = lb_two_servers(horizon=8)
    plan = lower(payload)
    ov = [("gen_users_mean", 0, 150.0), ("edge_mean", 2, 0.02), ("edge_dropout", 0, 0.2), ("step_time", 1, 0.03),
          ("gen_rpm_mean", 0, 35.0), ("edge_sigma", 1, 0.5)]
    ref_plan = lower(payload)
    ol.apply_overrides(ref_plan, {(k, i): v for k, i, v in ov})
    a = ol.simulate(ref_plan, 9)
    counts, clock, samples = hc.simulate(plan, 9, overrides=ov)
    assert np.array_equal(a.counts[:5].astype(np.uint32), counts[:5])
    assert np.array_equal(a.clock, clock) and np.array_equal(a.samples, samples)
    base = ol.simulate(plan, 9)
    assert not np.array_equal(base.counts[:3], a.counts[:3])


def test_capacity_overflow_is_flagged_never_silent():
    plan = lower(overload(horizon=8))
    counts, _, _ = hc.simulate(plan, 5, cap=16, fcap=8)
    assert int(counts[_abi.CNT_FLAGS]) & (_abi.FLAG_POOL_OVERFLOW | _abi.FLAG_FIFO_OVERFLOW)
    counts, _, _ = hc.simulate(lower(lb_two_servers(horizon=8)), 5, clock_capacity=10)
    assert int(counts[_abi.CNT_FLAGS]) & _abi.FLAG_CLOCK_OVERFLOW


def test_waiting_ram_puts_follow_simpy():
    """Fractional RAM needs (not multiples of 1/256 MB): simpy refuses some puts by one rounding and the response waits for
    the next RAM get of the server (af_core.hpp::m_srv_finish / m_put_trigger; such plans run every request event through
    the SimPy-order path).  The core against the oracle, which tests/test_reference_live.py holds to the live reference on
    this very family; single pass and lean-then-faithful."""
    from oracle.scenarios import fractional_ram_fuzz

    L = hc.lib()
    waits = dead = 0
    try:
        for case in range(48):
            plan = lower(fractional_ram_fuzz(random.Random(424200 + case), horizon=10))
            a = ol.simulate(plan, 900 + case)
            waits += a.put_waits > 0
            dead += (int(a.counts[_abi.CNT_FLAGS]) & _abi.FLAG_RAM_STARVED) != 0
            for two_pass in (0, 1):
                L.hc_set_two_pass(two_pass)
                counts, clock, samples = hc.simulate(plan, 900 + case, cap=16384, fcap=16384)
                assert np.array_equal(a.counts[:5].astype(np.uint32), counts[:5]), case
                assert np.array_equal(a.clock, clock) and np.array_equal(a.samples, samples), case
                assert int(counts[_abi.CNT_FLAGS]) == int(a.counts[_abi.CNT_FLAGS]), case
    finally:
        L.hc_set_two_pass(0)
    assert waits >= 20 and dead >= 5


def test_wait_queues_hold_more_than_16384_waiters():
    """The reference's simpy queues have no bound (server.py:146-149, 210-227); rounds 1-5 packed a server's two queue states
    into one word and stopped at 16 384 waiters (an OverflowError for 18 saturated payloads of round 5's fuzz).  Round 6: a
    word per queue, up to 2^20 waiters.  300 arrivals/s into a 100-requests/s server for 120 s: a backlog of 23 000."""
    from oracle.scenarios import _endpoint, _server, single_server

    for steps in ([("initial_parsing", 0.01), ("ram", 64), ("io_wait", 0.02)],      # the backlog waits for RAM (32 fit at once) ...
                  [("initial_parsing", 0.01), ("io_wait", 0.02)]):                  # ... for the core: the ready queue's series shows it
        payload = single_server(users=300, rpm=60, horizon=120, period=0.05)
        payload["topology_graph"]["nodes"]["servers"] = [_server("srv-1", 1, 2048, [_endpoint("/a", steps)])]
        plan = lower(payload)
        a = ol.simulate(plan, 5)
        assert int(a.counts[_abi.CNT_MAX_LIVE]) > 20_000
        counts, clock, samples = hc.simulate(plan, 5, cap=4096, fcap=16384)
        assert int(counts[_abi.CNT_FLAGS]) & _abi.FLAG_FIFO_OVERFLOW                  # reported, never silent
        counts, clock, samples = hc.simulate(plan, 5, cap=4096, fcap=65536)
        assert int(counts[_abi.CNT_FLAGS]) & _abi.FATAL_FLAGS == 0 and int(counts[_abi.CNT_MAX_LIVE]) == int(a.counts[_abi.CNT_MAX_LIVE])
        assert np.array_equal(a.counts[:5].astype(np.uint32), counts[:5])
        assert np.array_equal(a.clock, clock) and np.array_equal(a.samples, samples)
        if len(steps) == 2:
            assert int(samples[plan.n_edges].max()) > 16_384                            # ready_queue_len of srv-1


def test_ram_starved_endpoint_blocks_like_the_reference():
    """A request needing more RAM than ram_mb blocks that server's RAM queue for good (SURVEY trap 6)."""
    payload = stress_mixed(30)
    payload["topology_graph"]["nodes"]["servers"][2]["endpoints"][1]["steps"][1]["step_operation"]["necessary_ram"] = 5000
    plan = lower(payload)
    a = ol.simulate(plan, 1)
    counts, clock, samples = hc.simulate(plan, 1)
    assert int(a.counts[_abi.CNT_FLAGS]) & _abi.FLAG_RAM_STARVED
    assert int(counts[_abi.CNT_FLAGS]) & _abi.FLAG_RAM_STARVED
    assert np.array_equal(a.counts[:5].astype(np.uint32), counts[:5])
    assert np.array_equal(a.clock, clock) and np.array_equal(a.samples, samples)


def test_negative_spike_residue_is_reported_where_the_reference_raises():
    """DESIGN section 2, documented deviation: `transit + spike < 0` (residue of overlapping spikes under a zero transit time)
    raises ValueError("Negative delay") in the reference (edge.py:107, simpy) and yields no results; the engine delivers
    at now + (transit + spike) and REPORTS the scenario (AF_FLAG_NEGATIVE_DELAY; the Python runner raises the same
    ValueError for it).  Oracle, next-event core and stage-parallel kernel agree on the flag; the live reference's
    ValueError is shown in tests/test_reference_live.py."""
    from oracle.scenarios import negative_spike_residue

    plan = lower(negative_spike_residue(horizon=10))
    a = ol.simulate(plan, 3)
    assert int(a.counts[_abi.CNT_FLAGS]) & _abi.FLAG_NEGATIVE_DELAY
    counts, clock, samples = hc.simulate(plan, 3)
    assert int(counts[_abi.CNT_FLAGS]) & _abi.FLAG_NEGATIVE_DELAY
    assert np.array_equal(a.counts[:5].astype(np.uint32), counts[:5]) and np.array_equal(a.clock, clock)
    res = hc.flow_simulate(plan, 3, ring_rows=256)
    assert res is not None and int(res[0][_abi.CNT_FLAGS]) & hc.FLOW_FALLBACK, "the stage-parallel kernel hands such a scenario back"
    # before the residue exists (the second spike ends after the horizon) nothing is reported
    early = lower(negative_spike_residue(horizon=6, shift=2.0))
    assert not int(ol.simulate(early, 3).counts[_abi.CNT_FLAGS]) & _abi.FLAG_NEGATIVE_DELAY
    assert not int(hc.simulate(early, 3)[0][_abi.CNT_FLAGS]) & _abi.FLAG_NEGATIVE_DELAY


def test_zero_users_produces_ticks_only():
    payload = lb_two_servers(users=0.0, horizon=6)
    plan = lower(payload)
    counts, clock, samples = hc.simulate(plan, 3)
    assert counts[_abi.CNT_GENERATED] == 0 and len(clock) == 0
    assert counts[_abi.CNT_TICKS] == plan.tick_count and not samples.any()


def test_draw_capacity_overflow_is_flagged():
    plan = lower(lb_two_servers(horizon=8))
    counts, _, _ = hc.simulate(plan, 5, draw_capacity=100)
    assert int(counts[_abi.CNT_FLAGS]) & _abi.FLAG_DRAW_OVERFLOW
    assert int(counts[_abi.CNT_GENERATED]) == 100            # the arrival stream stops at the capacity
    counts, _, _ = hc.simulate(plan, 5)
    assert not int(counts[_abi.CNT_FLAGS]) & _abi.FLAG_DRAW_OVERFLOW
