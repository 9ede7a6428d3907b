"""The stage-parallel kernel (af_flow_kernel) on a real MI355X, through the C ABI.

* bit-identical to the oracle on the BASELINE workloads at their FULL sizes (T = 600 s; grid corners of
  configs 3 / 4, config 4 = sweep columns AND injected events, config 5 through both kernels and both
  state placements of the next-event kernel);
* bit-identical to the next-event kernels (`flow=False`) on whole batches, including scenarios it hands
  back (ties, RAM pressure, overflowing lists / rings): the hand-over is invisible in the results;
* statistical parity with the STOCK numpy-seeded reference from a committed fixture
  (tests/golden/pooled_lb2_numpy.npz, oracle/make_golden.py --pooled 256): SURVEY 8d criterion (2).
"""

from __future__ import annotations

import copy
import random

import numpy as np
import pytest

from asyncflow_amd import _abi
from asyncflow_amd.plan import lower
from asyncflow_amd.workloads import (
    BASELINE_SEED_BASE,
    fanout8,
    grid_users_rtt,
    lb_two_servers,
    lb_with_events,
    single_server,
    single_server_with_spike,
)
from oracle import oracle_lib as ol
from oracle.scenarios import flow_payload
from tests.conftest import GOLDEN_DIR

pytestmark = pytest.mark.gpu


def _runner(payload, **kw):
    from asyncflow_amd.runner import SimulationRunner

    return SimulationRunner(simulation_input=payload, **kw)


def _assert_scenario(got, want, what=""):
    assert np.array_equal(got.counts[:5].astype(np.uint64), want.counts[:5]), (what, got.counts, want.counts)
    assert int(got.counts[_abi.CNT_MARKS]) == int(want.counts[_abi.CNT_MARKS]), what
    assert np.array_equal(got.rqs_clock.view(np.uint64), want.clock.view(np.uint64)), f"{what}: rqs_clock differs"
    assert np.array_equal(got._samples, want.samples), f"{what}: sampled series differ"  # noqa: SLF001


def _same_batches(a, b):
    assert np.array_equal(a.counts[:, :6], b.counts[:, :6])          # generated .. ticks, flags
    assert np.array_equal(a.counts[:, _abi.CNT_MARKS], b.counts[:, _abi.CNT_MARKS])
    for i in range(len(a)):
        assert np.array_equal(a[i].rqs_clock.view(np.uint64), b[i].rqs_clock.view(np.uint64)), f"scenario {i}: rqs_clock"
        assert np.array_equal(a[i]._samples, b[i]._samples), f"scenario {i}: samples"  # noqa: SLF001


# ------------------------------------------------------------------------------------------------ batches
def test_lb2_batch_runs_on_the_flow_kernel_and_matches_the_oracle():
    payload = lb_two_servers(horizon=30)
    seeds = 0x5EED0000 + np.arange(200, dtype=np.uint64)
    res = _runner(payload, seeds=seeds).run()
    st = res.engine_stats
    assert st.flow_scenarios == 200 and st.flow_fallback == 0 and st.flow_list_entries in (64, 128) and st.flow_ring_rows >= 32
    plan = lower(payload)
    for i in (0, 1, 63, 64, 127, 199):
        _assert_scenario(res[i], ol.simulate(plan, int(seeds[i])), f"scenario {i}")
    want_counts = np.array([ol.simulate(plan, int(s), want_clock=False, want_samples=False).counts[:5] for s in seeds])
    assert np.array_equal(res.counts[:, :5].astype(np.uint64), want_counts)
    _same_batches(res, _runner(payload, seeds=seeds, flow=False).run())


@pytest.mark.parametrize("ring_rows", [0, 16, _abi.FLOW_RING_IN_HBM])
def test_tick_ring_placement_does_not_change_results(ring_rows):
    """auto / a ring far too small for some intervals (those scenarios are handed back) / differences in HBM."""
    payload = lb_with_events(users=200, horizon=40, scale=0.05)
    seeds = np.arange(96, dtype=np.uint64) + 31
    res = _runner(payload, seeds=seeds, flow_ring_rows=ring_rows).run()
    st = res.engine_stats
    assert st.flow_scenarios == 96
    if ring_rows == _abi.FLOW_RING_IN_HBM:
        assert st.flow_ring_rows == 0 and st.flow_fallback == 0
    assert st.flow_to_next_event == 0
    _same_batches(res, _runner(payload, seeds=seeds, flow=False).run())


def test_fanout_uses_larger_lists_and_far_edges():
    payload = fanout8(horizon=60)
    seeds = BASELINE_SEED_BASE[5] + np.arange(40, dtype=np.uint64)
    res = _runner(payload, seeds=seeds).run()
    st = res.engine_stats
    # ~1-s hops on an LDS ring of 1.6 s: the deliveries are entered by the receiving stations (FEAT_FAR), nothing is handed back
    assert st.flow_scenarios == 40 and st.flow_list_entries >= 128 and 16 <= st.flow_ring_rows <= 64 and st.flow_fallback == 0
    plan = lower(payload)
    for i in (0, 17, 39):
        _assert_scenario(res[i], ol.simulate(plan, int(seeds[i])), f"scenario {i}")
    small = _runner(payload, seeds=seeds, flow_list_entries=64).run()     # ~40 messages in flight per edge: lists overflow
    st = small.engine_stats
    assert st.flow_fallback_list > 0 and st.flow_retried == st.flow_fallback and st.flow_to_next_event == 0   # second chance: 256-entry lists
    _same_batches(res, small)
    big = _runner(payload, seeds=seeds, flow_list_entries=256).run()      # the 4-entries-per-lane instantiation
    assert big.engine_stats.flow_list_entries == 256 and big.engine_stats.flow_fallback == 0
    _same_batches(res, big)
    ring = _runner(payload, seeds=seeds, flow_ring_rows=16).run()         # 0.8 s of LDS ring against ~1-s hops: still nothing handed back
    assert ring.engine_stats.flow_ring_rows == 16 and ring.engine_stats.flow_fallback == 0
    _same_batches(res, ring)
    hbm = _runner(payload, seeds=seeds, flow_ring_rows=_abi.FLOW_RING_IN_HBM).run()
    assert hbm.engine_stats.flow_ring_rows == 0 and hbm.engine_stats.flow_fallback == 0
    _same_batches(res, hbm)


def test_a_ring_shorter_than_the_stay_in_a_server_hands_back():
    """What a ring still has to reach: the time a request spends inside its server (100-ms I/O step, 4 rows of 50 ms)."""
    payload = single_server(horizon=30)
    seeds = np.arange(48, dtype=np.uint64) + 5
    res = _runner(payload, seeds=seeds, flow_ring_rows=4).run()
    st = res.engine_stats
    assert st.flow_fallback_ring > 0 and st.flow_retried == st.flow_fallback and st.flow_to_next_event == 0   # second chance: differences in HBM
    _same_batches(res, _runner(payload, seeds=seeds).run())
    _assert_scenario(res[3], ol.simulate(lower(payload), int(seeds[3])), "scenario 3")


def test_a_handed_back_scenario_costs_a_wave_not_a_next_event_pass():
    """what the lean launch hands back is re-simulated by the second-chance launch of the same kernel -- one
    WAVE per scenario, 256-entry lists with send times, tick differences in HBM -- and only what THAT hands back goes to the
    next-event kernels (one lane per scenario, a latency-bound pass).  Eight LB-2 scenarios at the full 600-s horizon, forced
    off the lean launch by a 2-row tick ring: all eight come back, all eight are caught by the second chance, the two launches
    together take ~28 ms (the next-event kernels: ~970 ms; profiles/r04/handback_8_lb2_T600.json), bit-identical."""
    payload = lb_two_servers()
    seeds = 0x5EED0000 + np.arange(8, dtype=np.uint64)
    _runner(payload, seeds=seeds, flow_ring_rows=2).run()       # (warm: library load, first launches)
    res = _runner(payload, seeds=seeds, flow_ring_rows=2).run()
    st = res.engine_stats
    assert st.flow_fallback == 8 and st.flow_retried == 8 and st.flow_to_next_event == 0
    assert st.flow_kernel_ms < 60.0, st.flow_kernel_ms           # measured 27.7 ms; a next-event pass is 30 x that
    plan = lower(payload)
    for i in (0, 5):
        _assert_scenario(res[i], ol.simulate(plan, int(seeds[i])), f"scenario {i}")
    _same_batches(res, _runner(payload, seeds=seeds).run())


def test_handed_back_scenarios_are_invisible_in_the_results():
    """Fuzzed feed-forward payloads (idle to saturated, dyadic step times, tight RAM, spikes, outages): most
    batches contain scenarios the flow kernel hands back; results equal the next-event kernels' everywhere."""
    handed_back = ran = 0
    for case in range(24):
        rng = random.Random(31000 + case)
        payload = flow_payload(rng, horizon=6)
        seeds = np.arange(5, dtype=np.uint64) + 900 + case
        res = _runner(payload, seeds=seeds).run()
        st = res.engine_stats
        assert st.flow_scenarios == 5
        handed_back += st.flow_to_next_event
        ran += 5 - st.flow_to_next_event
        assert st.flow_retried <= st.flow_fallback and st.flow_to_next_event <= st.flow_fallback
        _same_batches(res, _runner(payload, seeds=seeds, flow=False).run())
        _assert_scenario(res[0], ol.simulate(lower(payload), int(seeds[0])), f"case {case}")
    assert handed_back >= 3 and ran > 20     # (the kernel models more every round: 4 of 120 are left, all exact ties)


def test_single_server_and_sweep_columns():
    res = _runner(single_server(horizon=120), seeds=[0, 1, 2]).run()
    assert res.engine_stats.flow_scenarios == 3 and res.engine_stats.flow_fallback == 0
    plan = lower(single_server(horizon=120))
    _assert_scenario(res[1], ol.simulate(plan, 1))
    base = lb_two_servers(horizon=30)
    users = np.array([20.0, 400.0, 900.0, 60.0])
    hop = np.array([0.001, 0.004, 0.0005, 0.03])
    cpu = np.array([0.002, 0.001, 0.0015, 0.004])
    sweep = {"rqs_input.avg_active_users.mean": users, "topology_graph.edges[*].latency.mean": hop,
             "topology_graph.nodes.servers[srv-1].endpoints[0].steps[0].cpu_time": cpu}
    seeds = np.arange(4, dtype=np.uint64) + 70
    res = _runner(base, seeds=seeds, sweep=sweep).run()
    assert res.engine_stats.flow_scenarios == 4
    for i in range(4):
        p = copy.deepcopy(base)
        p["rqs_input"]["avg_active_users"]["mean"] = float(users[i])
        for e in p["topology_graph"]["edges"]:
            e["latency"]["mean"] = float(hop[i])
        p["topology_graph"]["nodes"]["servers"][0]["endpoints"][0]["steps"][0]["step_operation"] = {"cpu_time": float(cpu[i])}
        _assert_scenario(res[i], ol.simulate(lower(p), int(seeds[i])), f"sweep point {i}")


def test_kernel_side_summary_and_no_outputs_modes():
    from oracle import analyzer_oracle as ao

    payload = lb_two_servers(horizon=30)
    seeds = np.arange(48, dtype=np.uint64) + 9
    bins, hist_max = 1024, 0.256
    both = _runner(payload, seeds=seeds, online_summary={"hist_bins": bins, "hist_max": hist_max}).run()
    assert both.engine_stats.flow_scenarios == 48 and both.engine_stats.flow_fallback == 0
    hist = both.online_hist.cpu().numpy().view(np.uint32)
    rps = both.online_rps.cpu().numpy()
    for i in (0, 47):
        ck = both[i].rqs_clock
        assert np.array_equal(hist[i], ao.latency_histogram(ck, bins, hist_max))
        assert np.array_equal(rps[i].astype(np.float64), ao.throughput_series(ck, 30)[1])
    lean = _runner(payload, seeds=seeds, collect_clock=False, collect_samples=False,
                   online_summary={"hist_bins": bins, "hist_max": hist_max}).run()
    assert np.array_equal(lean.counts[:, :5], both.counts[:, :5])
    assert np.array_equal(lean.online_hist.cpu().numpy(), both.online_hist.cpu().numpy())


# ------------------------------------------------------------------------- BASELINE configs at full size
def _grid_payload(base: dict, users: float, hop: float) -> dict:
    p = copy.deepcopy(base)
    p["rqs_input"]["avg_active_users"]["mean"] = users
    for e in p["topology_graph"]["edges"]:
        e["latency"]["mean"] = hop
    return p


@pytest.mark.parametrize("config", [3, 4])
def test_grid_corners_at_full_horizon(config):
    """users 10 / 1000 x per-hop latency 0.5 / 50 ms, T = 600 s, 2 seeds each (8 scenarios), as ONE sweep with
    per-scenario parameter columns; config 4 adds event_inj_lb.yml's spikes and outages.  Every scenario is
    compared with the oracle run on the payload that has the column values written into it."""
    base = lb_two_servers(horizon=600) if config == 3 else lb_with_events(users=400, horizon=600)
    users = np.array([10.0, 10.0, 1000.0, 1000.0] * 2)
    hop = np.array([0.0005, 0.05, 0.0005, 0.05] * 2)
    seeds = BASELINE_SEED_BASE[config] + np.arange(8, dtype=np.uint64)
    sweep = {"rqs_input.avg_active_users.mean": users, "topology_graph.edges[*].latency.mean": hop}
    res = _runner(base, seeds=seeds, sweep=sweep).run()
    st = res.engine_stats
    assert st.flow_scenarios == 8
    for i in range(8):
        want = ol.simulate(lower(_grid_payload(base, float(users[i]), float(hop[i]))), int(seeds[i]))
        _assert_scenario(res[i], want, f"config {config} users={users[i]} hop={hop[i]}")
    assert int(res.counts[:, _abi.CNT_TICKS].min()) == 11999
    seq = _runner(base, seeds=seeds, sweep=sweep, flow=False).run()           # the next-event kernels, same sweep
    _same_batches(res, seq)


def test_fanout_at_full_horizon_on_both_kernels_and_both_state_placements():
    payload = fanout8(horizon=600)
    seeds = BASELINE_SEED_BASE[5] + np.arange(8, dtype=np.uint64)
    plan = lower(payload)
    want = [ol.simulate(plan, int(s)) for s in seeds]
    flow = _runner(payload, seeds=seeds).run()
    assert flow.engine_stats.flow_scenarios == 8 and flow.engine_stats.flow_fallback == 0
    lds = _runner(payload, seeds=seeds, flow=False).run()                      # few scenarios -> narrow waves -> LDS state
    hbm = _runner(payload, seeds=seeds, flow=False, force_global_state=True).run()
    assert lds.engine_stats.state_in_lds == 1 and hbm.engine_stats.state_in_lds == 0
    for i in range(8):
        for name, res in (("flow", flow), ("lds", lds), ("hbm", hbm)):
            _assert_scenario(res[i], want[i], f"{name} scenario {i}")


# -------------------------------------------------------------------- statistical parity (criterion 2)
def test_statistical_parity_with_the_numpy_seeded_reference():
    """256 replicas of two_servers_lb.yml here vs 256 replicas of the STOCK reference (numpy PCG64 through the
    runner.rng seam; fixture written by `oracle/make_golden.py --pooled 256` in the build container):
    pooled p50 / p95 within 1 % and 3 standard errors, mean RPS within 1 %, means of the sampled ram_in_use and
    edge_concurrent_connection series within 2 %, two-sample KS at alpha = 0.01 on latencies thinned to one
    per ~7.5 simulated seconds (practically independent draws)."""
    fx = np.load(GOLDEN_DIR / "pooled_lb2_numpy.npz", allow_pickle=False)
    n = int(fx["n_seeds"])
    assert n >= 256
    seeds = 0x5EED0000 + np.arange(n, dtype=np.uint64)
    res = _runner(lb_two_servers(), seeds=seeds).run()
    assert res.engine_stats.flow_scenarios == n
    summ = res.summary(rps=True, series=True)
    stats = summ["stats"].cpu().numpy()
    ref = fx["stats"]
    for col, name, tol in ((2, "p50", 0.01), (4, "p95", 0.01), (1, "mean", 0.01), (5, "p99", 0.02)):
        a, b = stats[:, col], ref[:, col]
        se = np.sqrt(a.var(ddof=1) / len(a) + b.var(ddof=1) / len(b))
        assert abs(a.mean() - b.mean()) <= tol * b.mean(), (name, a.mean(), b.mean())
        assert abs(a.mean() - b.mean()) <= 3.0 * se + 1e-5 * b.mean(), (name, a.mean(), b.mean(), se)
    rps = summ["rps"].cpu().numpy().astype(np.float64).mean(axis=1)
    assert abs(rps.mean() - fx["rps_mean"].mean()) <= 0.01 * fx["rps_mean"].mean()
    # sampled series: means over ticks, averaged over replicas
    import json

    keys = json.loads(str(fx["series_keys"]))
    names = res.series_names()
    smean = summ["series_mean"].cpu().numpy()
    for j, key in enumerate(keys):
        metric, ent = key.split(":")
        if metric not in ("ram_in_use", "edge_concurrent_connection"):
            continue
        col = names.index(f"{ent}:{metric}")
        got, want = smean[:, col].mean(), fx["series_mean"][:, j].mean()
        assert abs(got - want) <= 0.02 * want, (key, got, want)
    # KS on thinned latencies: every 1000th completion of every replica, like the fixture
    thin = np.concatenate([(res[i].rqs_clock[500::1000, 1] - res[i].rqs_clock[500::1000, 0]) for i in range(n)])
    from scipy.stats import ks_2samp

    ks = ks_2samp(thin, fx["thin"])
    assert ks.pvalue > 0.01, (ks.statistic, ks.pvalue, len(thin), len(fx["thin"]))
    # pooled percentiles from the fixture's 10-us histogram vs ours
    lat = np.concatenate([res[i].rqs_clock[:, 1] - res[i].rqs_clock[:, 0] for i in range(0, n, 8)])
    h = fx["hist"].astype(np.float64)
    cdf = np.cumsum(h) / h.sum()
    width = float(fx["hist_max"]) / int(fx["hist_bins"])
    for q in (0.5, 0.95):
        ref_q = (np.searchsorted(cdf, q) + 0.5) * width
        assert abs(np.quantile(lat, q) - ref_q) <= 0.01 * ref_q, (q, np.quantile(lat, q), ref_q)


def test_grid_columns_helper_matches_survey_definition():
    a, b = grid_users_rtt(100)
    assert a.size == 10_000 and a.min() == 10.0 and a.max() == 1000.0 and b.min() == 0.0005 and abs(b.max() - 0.05) < 1e-15


def test_a_delivery_that_does_not_advance_the_clock_stays_on_the_flow_kernel():
    """Replica 101 670 of the BASELINE seed range draws an exponential transit time below half an ulp of the clock: the
    delivery happens at its own send instant.  That is an event of the NEXT station (it commutes with the sender's);
    round 2's first version handed the scenario back for it (1.1 s on a next-event kernel)."""
    seed = 0x5EED0000 + 101_670
    res = _runner(lb_two_servers(), seeds=[seed, seed + 1]).run()
    assert res.engine_stats.flow_scenarios == 2 and res.engine_stats.flow_fallback == 0
    _assert_scenario(res[0], ol.simulate(lower(lb_two_servers()), seed))
    _assert_scenario(res[1], ol.simulate(lower(lb_two_servers()), seed + 1))


def test_config2_replicas_at_full_horizon_match_the_oracle_bit_for_bit():
    """BASELINE config 2 at its real size per scenario (T = 600 s, ~76 000 completions, 11 999 ticks): replicas from the
    head, the middle and the end of the benched seed range 0x5EED0000 + [0, 10 000), every (start, finish) pair and every
    sample against the oracle, and scenario 0 against what the UNMODIFIED reference produced (tests/golden/lb2_rr_t600)."""
    picks = np.array([0, 1, 2, 3, 63, 64, 4095, 4096, 5000, 8191, 9998, 9999], dtype=np.uint64)
    seeds = BASELINE_SEED_BASE[2] + picks
    payload = lb_two_servers()
    res = _runner(payload, seeds=seeds).run()
    assert res.engine_stats.flow_scenarios == len(seeds) and res.engine_stats.flow_fallback == 0
    plan = lower(payload)
    for i, s in enumerate(seeds):
        _assert_scenario(res[i], ol.simulate(plan, int(s)), f"replica {int(picks[i])}")
    fx = np.load(GOLDEN_DIR / "lb2_rr_t600.npz", allow_pickle=False)
    assert int(fx["seed"]) == int(seeds[0])
    assert np.array_equal(res[0].rqs_clock, fx["clock"]) and np.array_equal(res[0]._samples, fx["samples"])  # noqa: SLF001


@pytest.mark.parametrize("per_wave", [0, 4, 5, 8, "groups", "groups64", "groups5"])
def test_arrival_pregeneration_group_widths_are_equivalent(monkeypatch, per_wave):
    """The arrival pre-generation as the engine launches it since round 3 (0: one DPP row of 16 lanes per scenario, the gaps
    of a batch handed along the row by `row_newbcast`) and in its round-2 forms (4, 5 or 8 scenarios per wave through the
    LDS crossbar): same arrival times whichever it is -- tiny sampling windows (window ends inside most batches),
    Gaussian users, a scenario count that leaves the last wave ragged.  "groups*": the form large sweeps of alike scenarios
    with long windows get (af_pregen.hpp, af_arrival_groups: one LANE per scenario for the sums and the window logic, the
    other waves of the workgroup work out the variates), one scenario per workgroup / all in one / ragged groups of five."""
    if str(per_wave).startswith("groups"):
        monkeypatch.setenv("AF_PREGEN_MODE", "groups")
        if per_wave != "groups":
            monkeypatch.setenv("AF_PREGEN_GROUP", per_wave[6:])
    elif per_wave:
        monkeypatch.setenv("AF_PREGEN_MODE", "rows")
        monkeypatch.setenv("AF_PREGEN_SCEN_PER_WAVE", str(per_wave))
    else:
        monkeypatch.setenv("AF_PREGEN_MODE", "rows")
    payload = lb_two_servers(horizon=40)
    payload["rqs_input"]["user_sampling_window"] = 1
    payload["rqs_input"]["avg_active_users"] = {"mean": 150, "distribution": "normal", "variance": 60}
    seeds = np.arange(37, dtype=np.uint64) + 5000
    res = _runner(payload, seeds=seeds).run()
    plan = lower(payload)
    for i in (0, 4, 5, 17, 36):
        _assert_scenario(res[i], ol.simulate(plan, int(seeds[i])), f"{per_wave} per wave, scenario {i}")
    want = np.array([ol.simulate(plan, int(s), want_clock=False, want_samples=False).counts[:5] for s in seeds])
    assert np.array_equal(res.counts[:, :5].astype(np.uint64), want)
    want_group = {"groups": 1, "groups64": 64, "groups5": 5}.get(per_wave, 0)
    assert res.engine_stats.pregen_group == want_group


@pytest.mark.parametrize("column", ["blocks", "alternating", "outliers"])
def test_grouped_pregeneration_on_sweeps_over_the_load(column):
    """A users column does not rule `af_arrival_groups` out: its time follows the HEAVIEST scenario of the launch whatever the
    others are (a workgroup's chain wave walks its heaviest scenario's draws, the lighter lanes idle), the row kernel's the total
    number of draws -- engine.hip, load_spread_suits_groups: grouped while heaviest <= 2.5 x mean (a grid written out
    users-major like BASELINE configs 3 / 4; two loads dealt out alternately), rows when a few scenarios are far heavier than
    the rest.  Either way the arrival times are the sequential sampler's: scenarios against the oracle run on the payload with
    the value written into it."""
    n = 3200
    if column == "blocks":
        users = np.repeat(np.linspace(60.0, 140.0, 32), 100)
    elif column == "alternating":
        users = np.where(np.arange(n) % 2 == 0, 40.0, 140.0)
    else:
        users = np.where(np.arange(n) % 400 == 7, 400.0, 20.0)
    base = lb_two_servers(horizon=20)
    seeds = 0x5EED0000 + np.arange(n, dtype=np.uint64)
    res = _runner(base, seeds=seeds, sweep={"rqs_input.avg_active_users.mean": users}).run()
    st = res.engine_stats
    assert st.flow_scenarios == n and st.flow_to_next_event == 0
    assert st.pregen_group == (0 if column == "outliers" else 13)   # the row kernel / ceil(3200 / 256) scenarios per workgroup
    for i in (0, 7, 99, 100, 1337, 3199):
        p = copy.deepcopy(base)
        p["rqs_input"]["avg_active_users"]["mean"] = float(users[i])
        _assert_scenario(res[i], ol.simulate(lower(p), int(seeds[i])), f"{column} scenario {i}")


def test_launch_order_of_a_sweep_over_the_load_does_not_change_results(monkeypatch):
    """Sweeps with a users / rpm column launch their heaviest scenarios first (engine.hip, heaviest_first: wave j simulates
    scenario order[j]; a users-ascending grid would otherwise end with a few long waves on an empty chip -- 86.0 -> 60.7 ms on
    the 100 x 100 grid of BASELINE config 3 written out users-major).  Outputs stay at the scenario's index: same batch with
    the order switched off, scenarios against the oracle."""
    n = 300
    rng = np.random.default_rng(11)
    users = rng.choice([15.0, 60.0, 240.0], size=n)
    base = lb_two_servers(horizon=15)
    seeds = 0x5EED0000 + np.arange(n, dtype=np.uint64)
    sweep = {"rqs_input.avg_active_users.mean": users}
    res = _runner(base, seeds=seeds, sweep=sweep).run()
    assert res.engine_stats.flow_scenarios == n
    monkeypatch.setenv("AF_FLOW_ORDER_OFF", "1")
    _same_batches(res, _runner(base, seeds=seeds, sweep=sweep).run())
    for i in (0, 1, 150, 299):
        p = copy.deepcopy(base)
        p["rqs_input"]["avg_active_users"]["mean"] = float(users[i])
        _assert_scenario(res[i], ol.simulate(lower(p), int(seeds[i])), f"scenario {i}")


# ------------------------------------------------------------------- seconds-long spikes (reference examples)
@pytest.mark.parametrize("heavy", [False, True])
def test_reference_spike_examples_stay_on_the_flow_kernel(heavy):
    """examples/yaml_input/data/event_inj_single_server.yml and heavy_inj_single_server.yml at their full length:
    a 2 s / 3 s spike on the client -> server edge.  The station behind the spike runs ahead of the client while it
    lasts; when it ends, rate x spike messages wait at the servers -- the long-list instantiation carries them
    (heavy: the whole launch starts there; light: the few scenarios that outgrow 64 entries get it as second chance)."""
    payload = single_server_with_spike(heavy=heavy)
    seeds = 0x5EED0000 + np.arange(96, dtype=np.uint64)
    res = _runner(payload, seeds=seeds).run()
    st = res.engine_stats
    assert st.flow_scenarios == 96 and st.flow_to_next_event == 0, (st.flow_fallback, st.flow_fallback_list, st.flow_fallback_tie, st.flow_fallback_ram)
    if heavy:
        assert st.flow_list_entries >= 512 and st.flow_fallback == 0
    plan = lower(payload)
    for i in (0, 2, 95):
        _assert_scenario(res[i], ol.simulate(plan, int(seeds[i])), f"scenario {i}")
    _same_batches(res, _runner(payload, seeds=seeds, flow=False).run())


# ------------------------------------------------------------------------------------ least connections
def test_least_connections_runs_on_the_flow_kernel():
    """lb_algorithms.py:10-20: fewest messages in flight on the LB's out-edges -- a decision of the LB station alone."""
    payload = lb_two_servers(horizon=60, algo="least_connection")
    seeds = 0x1EA50000 + np.arange(160, dtype=np.uint64)
    res = _runner(payload, seeds=seeds).run()
    st = res.engine_stats
    assert st.flow_scenarios == 160 and st.flow_to_next_event == 0
    plan = lower(payload)
    for i in (0, 1, 159):
        _assert_scenario(res[i], ol.simulate(plan, int(seeds[i])), f"scenario {i}")
    _same_batches(res, _runner(payload, seeds=seeds, flow=False).run())
    # with injected outages (the candidate list changes) and spikes, on the 8-server fan-out too
    for payload in (lb_with_events(users=300, horizon=60, scale=0.1), fanout8(horizon=40)):
        payload["topology_graph"]["nodes"]["load_balancer"]["algorithms"] = "least_connection"
        seeds = np.arange(64, dtype=np.uint64) + 77
        res = _runner(payload, seeds=seeds).run()
        assert res.engine_stats.flow_scenarios == 64 and res.engine_stats.flow_to_next_event == 0
        _same_batches(res, _runner(payload, seeds=seeds, flow=False).run())


def test_plan_specialised_flow_kernel_gives_identical_results():
    """asyncflow_amd/jit.py: af_flow_kernel compiled once more with the plan's shape, the horizon / tick constants and the
    LDS layout as immediates (`af_flow_jit`), for every kind of instantiation the launcher picks: plain lean (LB-2),
    marks + far (injected events), 128-entry lists with far edges (fan-out), least connections, per-scenario parameter
    columns (the blob is still patched per scenario), no sampled series, kernel-side summary."""
    cases = [
        (lb_two_servers(horizon=30), {}),
        (lb_with_events(users=200, horizon=40, scale=0.05), {}),
        (fanout8(horizon=30), {}),
        (lb_two_servers(horizon=30, algo="least_connection"), {}),
        (single_server(horizon=40), {"collect_samples": False}),
        (lb_two_servers(horizon=20), {"online_summary": {"hist_bins": 512, "hist_max": 0.256}}),
        (lb_two_servers(horizon=20), {"sweep": {"rqs_input.avg_active_users.mean": np.linspace(20.0, 700.0, 70),
                                                "topology_graph.edges[*].latency.mean": np.linspace(0.0005, 0.02, 70)}}),
    ]
    for k, (payload, kw) in enumerate(cases):
        seeds = np.arange(70, dtype=np.uint64) + 31 + 100 * k
        generic = _runner(payload, seeds=seeds, specialise=False, **kw).run()
        special = _runner(payload, seeds=seeds, specialise=True, **kw).run()
        assert generic.engine_stats.specialised_launches == 0, k
        assert special.engine_stats.specialised_launches >= 1 and special.engine_stats.jit_fallbacks == 0, k
        assert special.engine_stats.flow_scenarios == 70
        assert np.array_equal(generic.counts, special.counts), k
        for i in (0, 33, 69):
            assert np.array_equal(generic[i].rqs_clock, special[i].rqs_clock), (k, i)
            if kw.get("collect_samples", True):
                assert np.array_equal(generic[i]._samples, special[i]._samples), (k, i)  # noqa: SLF001
        if "online_summary" in kw:
            assert np.array_equal(generic.online_hist.cpu().numpy(), special.online_hist.cpu().numpy())
        if "sweep" not in kw and kw.get("collect_samples", True):
            _assert_scenario(special[5], ol.simulate(lower(payload), int(seeds[5])), f"case {k}")


def test_prebuilt_kernels_serve_a_box_without_hipcc(tmp_path, monkeypatch):
    """The headline must not depend on a compiler where the bench runs: a planning-only engine
    (AF_DEVICE_PLAN_ONLY, no device) writes the same spec as the engine on the GPU, `bench.prebuild_kernels` compiles it
    ahead of time, and with hipcc out of reach the run still launches the plan-specialised kernel: no fallback."""
    import bench
    from asyncflow_amd import jit
    from asyncflow_amd.engine import PLAN_ONLY, Engine

    args = bench.make_parser().parse_args(["--config", "2"])
    args.horizon = None
    shape = bench.rank_shape(bench.build_workload(2, 0, 1, 0, None), args)
    kw = dict(clock_ptr=8, clock_capacity=shape["clock_cap"], samples_ptr=8, tick_capacity=shape["ticks"], counts_ptr=8,
              draw_capacity=shape["clock_cap"])
    specs = []
    for device in (PLAN_ONLY, 0):
        eng = Engine(shape["plan"], device, **shape["engine_kw"])
        specs.append(eng.jit_spec(shape["seeds"], [], **kw))
        eng.close()
    assert specs[0] == specs[1]
    monkeypatch.setattr(jit, "CACHE_DIR", tmp_path)
    monkeypatch.setattr(jit, "_FALLBACK_CACHE_DIR", tmp_path / "none")
    assert bench.prebuild_kernels(configs=(2,), worlds=(1,), verbose=False) == [specs[0]]
    monkeypatch.setenv("ASYNCFLOW_NO_HIPCC", "1")
    # a short run of the same plan shape + layout (the horizon is part of the spec: the full one, 256 replicas)
    res = _runner(lb_two_servers(), seeds=np.arange(256, dtype=np.uint64) + 0x5EED0000, specialise=True).run()
    st = res.engine_stats
    assert st.specialised_launches >= 1 and st.jit_fallbacks == 0 and st.flow_scenarios == 256
    _assert_scenario(res[255], ol.simulate(lower(lb_two_servers()), 0x5EED0000 + 255), "prebuilt kernel")


def test_far_and_near_lean_instantiations_agree(monkeypatch):
    """LB-2's hops are fast: the engine launches the plain lean instantiation (the sender enters both ends of every
    message); AF_FLOW_FORCE_FAR makes it launch the FEAT_FAR one with the window rule of slow-hop plans.  Same results."""
    payload = lb_two_servers(horizon=40)
    seeds = 0xFA50000 + np.arange(96, dtype=np.uint64)
    near = _runner(payload, seeds=seeds).run()
    monkeypatch.setenv("AF_FLOW_FORCE_FAR", "1")
    far = _runner(payload, seeds=seeds).run()
    assert near.engine_stats.flow_fallback == 0 and far.engine_stats.flow_fallback == 0
    _same_batches(near, far)
    _assert_scenario(far[5], ol.simulate(lower(payload), int(seeds[5])), "scenario 5")


# ---------------------------------------------------------------- sweeps over events / resources / window
def test_sweep_over_spike_size_and_outage_window_matches_the_oracle_per_point():
    """BASELINE config 4's topology and events (event_inj_lb.yml): a grid spike size x outage window x users, plus the
    server-resource and sampling-window axes; every scenario against the oracle run on the payload with that point's
    values WRITTEN INTO it (the reference's way of running the point), on both kernel families."""
    from asyncflow_amd.runner import write_point
    from asyncflow_amd.sweep import expand_grid

    base = lb_with_events(users=300, horizon=120, scale=0.2)
    ev = {e["event_id"]: e for e in lower(base).payload["events"]}
    t0, t1 = ev["ev-srv1-down"]["start"]["t_start"], ev["ev-srv1-down"]["end"]["t_end"]
    grid = expand_grid({"events[ev-spike-1].start.spike_s": [0.005, 0.02, 0.08],
                        "events[ev-srv1-down].end.t_end": [t1 - 5.0, t1, t1 + 10.0],
                        "rqs_input.avg_active_users.mean": [80.0, 300.0]}, replicas=2, seed_base=0xC0F40000)
    n = len(grid)
    cols = dict(grid.columns)
    cols["events[ev-srv1-down].start.t_start"] = np.where(cols["events[ev-srv1-down].end.t_end"] > t1, t0 + 4.0, t0)
    cols["topology_graph.nodes.servers[srv-1].server_resources.cpu_cores"] = np.tile([1.0, 2.0, 3.0], n // 3)
    cols["topology_graph.nodes.servers[srv-2].server_resources.ram_mb"] = np.tile([512.0, 2048.0], n // 2)
    cols["rqs_input.user_sampling_window"] = np.tile([5.0, 60.0, 20.0, 1.0], n // 4)
    res = _runner(base, seeds=grid.seeds, sweep=cols).run()
    st = res.engine_stats
    assert st.flow_scenarios == n and st.flow_to_next_event <= 2
    plan = lower(base)
    for i in range(n):
        point = copy.deepcopy(plan.payload)
        for key, col in cols.items():
            write_point(point, key, col[i])
        _assert_scenario(res[i], ol.simulate(lower(point), int(grid.seeds[i])), f"point {i}")
    _same_batches(res, _runner(base, seeds=grid.seeds, sweep=cols, flow=False).run())
    special = _runner(base, seeds=grid.seeds, sweep=cols, specialise=True).run()     # the plan-specialised build patches the same blob
    assert special.engine_stats.specialised_launches >= 1
    _same_batches(res, special)


def test_a_sweep_the_flow_kernel_cannot_be_sized_for_runs_on_the_next_event_kernels():
    """a cpu_cores column above the stage-parallel kernel's 64 cores per server used to fail the whole run with
    AF_ERR_CAPACITY, while a PLAN with that many cores simply ran on the next-event kernels.  Now the sweep does too
    (and the runner's spec query with it); flow="always" still says why it cannot."""
    from asyncflow_amd.engine import EngineError
    from asyncflow_amd.runner import write_point

    base = lb_two_servers(horizon=20)
    key = "topology_graph.nodes.servers[srv-1].server_resources.cpu_cores"
    cols = {key: np.array([1.0, 2.0, 80.0, 65.0])}
    seeds = np.arange(4, dtype=np.uint64) + 11
    res = _runner(base, seeds=seeds, sweep=cols).run()
    assert res.engine_stats.flow_scenarios == 0
    plan = lower(base)
    for i in range(4):
        point = copy.deepcopy(plan.payload)
        write_point(point, key, cols[key][i])
        _assert_scenario(res[i], ol.simulate(lower(point), int(seeds[i])), f"point {i}")
    with pytest.raises(EngineError, match="64 cores"):
        _runner(base, seeds=seeds, sweep=cols, flow="always").run()


def test_negative_delay_raises_like_the_reference_or_is_flagged():
    """transit + spike < 0 (negative residue of overlapping spikes under a zero transit time): the reference raises
    ValueError("Negative delay") (edge.py:107); so does the runner -- or, with on_negative_delay="flag", the scenario keeps
    its results (equal to the oracle's) and carries AF_FLAG_NEGATIVE_DELAY.  The stage-parallel kernel hands such scenarios
    to the next-event kernels, which report them."""
    from oracle.scenarios import negative_spike_residue

    payload = negative_spike_residue(horizon=10)
    seeds = np.arange(6, dtype=np.uint64) + 3
    with pytest.raises(ValueError, match="Negative delay"):
        _runner(payload, seeds=seeds).run()
    res = _runner(payload, seeds=seeds, on_negative_delay="flag").run()
    assert res.engine_stats.flow_scenarios == 6 and res.engine_stats.flow_to_next_event >= 1
    plan = lower(payload)
    flagged = 0
    for i, s in enumerate(seeds):
        want = ol.simulate(plan, int(s))
        _assert_scenario(res[i], want, f"scenario {i}")
        assert (int(res.flags[i]) & _abi.FLAG_NEGATIVE_DELAY) == (int(want.counts[_abi.CNT_FLAGS]) & _abi.FLAG_NEGATIVE_DELAY)
        flagged += bool(int(res.flags[i]) & _abi.FLAG_NEGATIVE_DELAY)
    assert flagged >= 1


# ------------------------------------------------------------------------- round 3: f3 stragglers on the fast path
def test_poisson_latencies_and_wide_round_robin_fan_out_run_on_the_flow_kernel():
    """Poisson (whole-second, zero included) edge latencies and 9..12 servers behind a round-robin LB are inside the
    stage-parallel kernel's range since round 3: bit-identical to the oracle and to the next-event kernels, hand-backs
    (lists that overflow under 1-s hops, the odd genuine tie) invisible in the results."""
    from oracle.scenarios import wide_fanout

    p = lb_two_servers(horizon=40, users=60)
    for e in p["topology_graph"]["edges"]:
        if e["id"] in ("client-lb", "srv1-client"):
            e["latency"] = {"mean": 0.3, "distribution": "poisson"}
    seeds = np.arange(48, dtype=np.uint64) + 700
    res = _runner(p, seeds=seeds).run()
    st = res.engine_stats
    assert res.flow_reason == "" and st.flow_scenarios == 48 and st.flow_to_next_event <= 4
    plan = lower(p)
    for i in (0, 11, 47):
        _assert_scenario(res[i], ol.simulate(plan, int(seeds[i])), f"poisson scenario {i}")
    _same_batches(res, _runner(p, seeds=seeds, flow=False).run())

    wide = wide_fanout(10, "round_robin", horizon=20, users=100)
    for s in wide["topology_graph"]["nodes"]["servers"]:
        s["endpoints"] = s["endpoints"][:1]
    seeds = np.arange(32, dtype=np.uint64) + 70
    res = _runner(wide, seeds=seeds).run()
    assert res.flow_reason == "" and res.engine_stats.flow_scenarios == 32
    plan = lower(wide)
    for i in (0, 31):
        _assert_scenario(res[i], ol.simulate(plan, int(seeds[i])), f"10-server scenario {i}")
    _same_batches(res, _runner(wide, seeds=seeds, flow=False).run())
    special = _runner(wide, seeds=seeds, specialise=True).run()
    assert special.engine_stats.specialised_launches >= 1
    _same_batches(res, special)


@pytest.mark.parametrize("n_srv", [9, 16])
def test_least_connections_over_nine_to_sixteen_servers_run_on_the_flow_kernel(n_srv):
    """Round 6: least connections in front of 9 .. 16 servers is inside the stage-parallel kernel's range (Flow::lb_pick_lc_n<16>:
    sixteen 16-bit in-flight counts in four words, sixteen prepared draws per lane; rounds 2-5: eight, wider fan-outs went to the
    next-event kernels).  Against the oracle, against the next-event kernels, generic and plan-specialised builds."""
    from oracle.scenarios import wide_fanout

    wide = wide_fanout(n_srv, "least_connection", horizon=16, users=100)
    for s in wide["topology_graph"]["nodes"]["servers"]:
        s["endpoints"] = s["endpoints"][:1]
    seeds = np.arange(24, dtype=np.uint64) + 170 + n_srv
    res = _runner(wide, seeds=seeds).run()
    assert res.flow_reason == "" and res.engine_stats.flow_scenarios == 24, res.flow_reason
    plan = lower(wide)
    for i in (0, 9, 23):
        _assert_scenario(res[i], ol.simulate(plan, int(seeds[i])), f"{n_srv}-server least-connections scenario {i}")
    _same_batches(res, _runner(wide, seeds=seeds, flow=False).run())
    special = _runner(wide, seeds=seeds, specialise=True).run()
    assert special.engine_stats.specialised_launches >= 1
    _same_batches(res, special)


def test_servers_that_feed_servers_run_on_the_flow_kernel():
    """Round 4: server -> server edges are inside the stage-parallel kernel's range (FEAT_CHAIN: the servers in
    levels, the server station once per level and round over the one server list, each level with its own horizon).
    Fuzzed tiers -- [LB ->] front servers -> [middle ->] backend -> client, up to three levels, spikes and outages -- against the
    oracle and against the next-event kernels, generic and plan-specialised builds; what the lean launch hands back (second-long
    log-normal hops outgrow its lists) is caught by the second-chance launch, itself a FEAT_CHAIN instantiation."""
    from oracle.scenarios import server_tiers

    stayed = total = 0
    for k in range(10):
        payload = server_tiers(random.Random(91000 + k), horizon=12)
        seeds = np.arange(8, dtype=np.uint64) + 50 * k
        res = _runner(payload, seeds=seeds).run()
        st = res.engine_stats
        assert res.flow_reason == "" and st.flow_scenarios == 8, res.flow_reason
        stayed += 8 - st.flow_to_next_event
        total += 8
        plan = lower(payload)
        for i in (0, 7):
            _assert_scenario(res[i], ol.simulate(plan, int(seeds[i])), f"tiers {k} scenario {i}")
        _same_batches(res, _runner(payload, seeds=seeds, flow=False).run())
        if k < 3:
            special = _runner(payload, seeds=seeds, specialise=True).run()
            assert special.engine_stats.specialised_launches >= 1
            _same_batches(res, special)
    assert stayed >= total - 4, f"{stayed} of {total} scenarios stayed on the stage-parallel kernel"


def test_server_tiers_behind_a_least_connections_lb_run_on_the_flow_kernel():
    """Round 4: server tiers behind a least-connections LB (FEAT_CHAIN | FEAT_LC): the walk counts the list entries that came by
    the LB's own edges.  Against the oracle and the next-event kernels."""
    from oracle.scenarios import server_tiers

    ran = 0
    for k in range(16):   # (the last four: general servers in the tiers -- FEAT_GENSRV | FEAT_LC | FEAT_CHAIN)
        payload = server_tiers(random.Random(94000 + k), horizon=12, algo="least_connection", general=k >= 12)
        if "load_balancer" not in payload["topology_graph"]["nodes"]:
            continue
        ran += 1
        seeds = np.arange(8, dtype=np.uint64) + 90 * k
        res = _runner(payload, seeds=seeds).run()
        assert res.flow_reason == "" and res.engine_stats.flow_scenarios == 8, res.flow_reason
        plan = lower(payload)
        for i in (0, 7):
            _assert_scenario(res[i], ol.simulate(plan, int(seeds[i])), f"least-connections tiers {k} scenario {i}")
        _same_batches(res, _runner(payload, seeds=seeds, flow=False).run())
    assert ran >= 4


def test_tiers_of_general_servers_run_on_the_flow_kernel():
    """Round 4: server tiers whose servers have two endpoints or come back to the core queue after an I/O step
    (FEAT_GENSRV | FEAT_CHAIN: the event-by-event station runs the servers of each level up to that level's horizon) against
    the oracle and the next-event kernels; first launch in the compact form, hand-backs to the second-chance form."""
    from oracle.scenarios import server_tiers

    stayed = total = 0
    for k in range(8):
        payload = server_tiers(random.Random(93000 + k), horizon=12, general=True)
        seeds = np.arange(8, dtype=np.uint64) + 70 * k
        res = _runner(payload, seeds=seeds).run()
        st = res.engine_stats
        assert res.flow_reason == "" and st.flow_scenarios == 8, res.flow_reason
        stayed += 8 - st.flow_to_next_event
        total += 8
        plan = lower(payload)
        for i in (0, 7):
            _assert_scenario(res[i], ol.simulate(plan, int(seeds[i])), f"general tiers {k} scenario {i}")
        _same_batches(res, _runner(payload, seeds=seeds, flow=False).run())
    assert stayed >= total - 4, f"{stayed} of {total} scenarios stayed on the stage-parallel kernel"


def test_several_endpoints_per_server_run_on_the_flow_kernel():
    """Round 3: plans whose servers have several endpoints, come back to the core queue after an I/O step or need
    different amounts of RAM per request run on the stage-parallel kernel (its server station simulates each server event by
    event, Flow::gen_servers); scenarios it hands back (two events of one server at one instant, > 32 requests inside one
    server) are invisible in the results."""
    from oracle.scenarios import overload, random_payload, stress_mixed, wide_fanout

    stayed = total = 0
    for k, payload in enumerate([stress_mixed(40), overload(20), wide_fanout(8, "round_robin", horizon=12, users=100)]
                                + [random_payload(random.Random(31000 + c), horizon=8) for c in range(12)]):
        seeds = np.arange(6, dtype=np.uint64) + 40 * k
        res = _runner(payload, seeds=seeds, on_negative_delay="flag").run()
        st = res.engine_stats
        assert res.flow_reason == "" and st.flow_scenarios == 6, res.flow_reason
        stayed += 6 - st.flow_to_next_event
        total += 6
        plan = lower(payload)
        for i in (0, 5):
            _assert_scenario(res[i], ol.simulate(plan, int(seeds[i])), f"payload {k} scenario {i}")
        _same_batches(res, _runner(payload, seeds=seeds, flow=False, on_negative_delay="flag").run())
    assert stayed >= total // 2, f"{stayed} of {total} scenarios stayed on the stage-parallel kernel"
    two_ep = wide_fanout(8, "round_robin", horizon=12, users=100)
    seeds = np.arange(24, dtype=np.uint64) + 900
    generic = _runner(two_ep, seeds=seeds, specialise=False, flow="always").run()
    special = _runner(two_ep, seeds=seeds, specialise=True, flow="always").run()     # (launches above 64 KB of LDS per wave keep the generic build)
    assert generic.engine_stats.flow_scenarios == 24
    _same_batches(generic, special)
    # the engine's own choice (flow=True) is the stage-parallel kernel at every sweep size since round 4 (shared instants are
    # resolved in the station, the first launch needs 17.6 instead of 41 KB of LDS per wave: profiles/r04/gensrv_*.json)
    auto = _runner(two_ep, seeds=seeds).run()
    assert auto.engine_stats.flow_scenarios == 24 and auto.flow_reason == ""
    _same_batches(generic, auto)


def test_round_step_times_share_instants_inside_a_general_server():
    """LB-2 with a second endpoint whose steps are round numbers of milliseconds: step ends of different requests of one server
    coincide as soon as requests queue for the core.  The station runs those instants the way SimPy does (Flow::gs_instant:
    the instant's Timeouts in creation order, then one FIFO of the zero-time steps they schedule) instead of handing the
    scenario back: all 64 scenarios stay at T = 120 s (round 3: 5.6 % came back at 120 s, 42 % at 600 s), on the compact
    first-launch form the engine picks by itself, and every result equals the next-event kernels' and the oracle's."""
    from asyncflow_amd.workloads import _endpoint

    p = lb_two_servers(horizon=120)
    for s in p["topology_graph"]["nodes"]["servers"]:
        s["endpoints"].append(_endpoint("/report", [("io_db", 0.004), ("ram", 64), ("cpu_bound_operation", 0.0015),
                                                     ("io_wait", 0.006), ("cpu_bound_operation", 0.0005)]))
    seeds = 0x5EED0000 + np.arange(64, dtype=np.uint64)
    res = _runner(p, seeds=seeds).run()
    st = res.engine_stats
    assert st.flow_scenarios == 64 and st.flow_fallback == 0 and st.flow_to_next_event == 0, (st.flow_scenarios, st.flow_fallback)
    plan = lower(p)
    for i in (0, 7, 19, 63):
        _assert_scenario(res[i], ol.simulate(plan, int(seeds[i])), f"scenario {i}")
    _same_batches(res, _runner(p, seeds=seeds, flow=False).run())


# ------------------------------------------------------------------------------------------ round 5
def test_general_servers_are_solved_a_round_at_a_time():
    """Round 5: the general server station solves a whole round at once (Flow::gen_servers_par: a lane is a
    request, the FIFO recurrence of a server with up to four cores relaxed to its fixed point) and leaves to the event-by-event walk only what it cannot
    decide.  Two-endpoint LB-2 -- idle to loaded -- on the generic and the plan-specialised build: nearly every round is solved
    at once (the kernel reports rounds solved << 16 | rounds walked in counts[:, CNT_MAX_LIVE]), nothing is handed back, and
    every result equals the next-event kernels' and the oracle's."""
    from asyncflow_amd.workloads import lb_two_servers_two_endpoints

    for users, n, T, cores in ((400, 96, 60, (1, 1)), (1500, 32, 20, (1, 1)), (3000, 32, 20, (2, 3)), (5000, 16, 15, (4, 2))):
        p = lb_two_servers_two_endpoints(users=users, horizon=T)
        for srv, c in zip(p["topology_graph"]["nodes"]["servers"], cores):      # (c cores: the c-th latest release among the predecessors)
            srv["server_resources"]["cpu_cores"] = c
        seeds = 0x5EED0000 + np.arange(n, dtype=np.uint64)
        res = _runner(p, seeds=seeds, specialise=False).run()
        st = res.engine_stats
        # (servers near saturation -- the last two cases -- may see a genuine tie or more than 32 requests inside: handed back, exact)
        assert st.flow_scenarios == n and st.flow_to_next_event <= (0 if users <= 1500 else 3), (users, st.flow_scenarios, st.flow_fallback)
        rounds = res.counts[:, _abi.CNT_MAX_LIVE].astype(np.int64)
        at_once, walked = int((rounds >> 16).sum()), int((rounds & 0xFFFF).sum())
        assert at_once > (8 if users <= 3000 else 2) * walked, (users, at_once, walked)
        plan = lower(p)
        for i in (0, n // 2, n - 1):
            _assert_scenario(res[i], ol.simulate(plan, int(seeds[i])), f"users {users} scenario {i}")
        _same_batches(res, _runner(p, seeds=seeds, flow=False).run())
        special = _runner(p, seeds=seeds, specialise=True).run()
        assert special.engine_stats.specialised_launches >= 1
        _same_batches(res, special)


def test_the_general_server_benchmark_batch_is_identical_on_both_kernel_families():
    """`bench.py --config 6` (two-endpoint LB-2) at 2 048 replicas x 600 s: every scenario of the batch, on the device."""
    import torch

    from asyncflow_amd.results import differing_scenarios
    from tests.test_gpu_full_batches import _oracle_bulk, _sweep

    flow = _sweep(6, 2048, [])
    acc = flow.step()
    torch.cuda.synchronize()
    assert acc["flow_scen"] == flow.n == 2048 and acc["flow_fallback"][0] == 0
    assert _oracle_bulk(flow, 256) == 256          # (round 5: 8 picks) every eighth scenario against the oracle itself
    seq = _sweep(6, 2048, ["--no-flow", "--generic-kernels"])
    seq.step()
    torch.cuda.synchronize()
    differ = differing_scenarios(flow.counts, flow.clock, flow.samples, seq.counts, seq.clock, seq.samples)
    assert differ.size == 0, differ[:8]
    flow.eng.close()
    seq.eng.close()


@pytest.mark.parametrize("n_srv", [13, 16])
def test_thirteen_to_sixteen_servers_behind_a_round_robin_lb(n_srv):
    """Round 5: 2 + 5 S sampled series are more than a wave has lanes for S > 12; a lane then carries the
    running values of two series (Flow::flush_ticks).  Generic and plan-specialised builds against the next-event kernels and
    the oracle."""
    from oracle.scenarios import wide_fanout

    wide = wide_fanout(n_srv, "round_robin", horizon=20, users=100)
    for s in wide["topology_graph"]["nodes"]["servers"]:
        s["endpoints"] = s["endpoints"][:1]
    seeds = np.arange(24, dtype=np.uint64) + 170
    res = _runner(wide, seeds=seeds).run()
    assert res.flow_reason == "" and res.engine_stats.flow_scenarios == 24, res.flow_reason
    plan = lower(wide)
    for i in (0, 23):
        _assert_scenario(res[i], ol.simulate(plan, int(seeds[i])), f"{n_srv}-server scenario {i}")
    _same_batches(res, _runner(wide, seeds=seeds, flow=False).run())
    special = _runner(wide, seeds=seeds, specialise=True).run()
    assert special.engine_stats.specialised_launches >= 1
    _same_batches(res, special)


@pytest.mark.parametrize(("depth", "fan"), [(4, True), (5, True), (5, False)])
def test_server_chains_of_four_and_five_levels_run_on_the_flow_kernel(depth, fan):
    """Round 5: tiers deeper than three levels; the generic build (horizons in LDS: no scratch object) and
    the plan-specialised build (as many unrolled level passes as the plan has levels) against the next-event kernels."""
    from oracle.scenarios import deep_chain

    payload = deep_chain(depth, users=150, horizon=30, fan=fan)
    seeds = np.arange(32, dtype=np.uint64) + 11 * depth
    res = _runner(payload, seeds=seeds, specialise=False).run()
    assert res.flow_reason == "" and res.engine_stats.flow_scenarios == 32 and res.engine_stats.flow_to_next_event <= 2, res.flow_reason
    plan = lower(payload)
    for i in (0, 31):
        _assert_scenario(res[i], ol.simulate(plan, int(seeds[i])), f"depth {depth} scenario {i}")
    _same_batches(res, _runner(payload, seeds=seeds, flow=False).run())
    special = _runner(payload, seeds=seeds, specialise=True).run()
    assert special.engine_stats.specialised_launches >= 1
    _same_batches(res, special)


def test_a_short_sweep_gets_its_specialised_kernel_from_the_second_run_on(tmp_path, monkeypatch):
    """`specialise=None` (the default): a sweep too short to repay a hipcc run is simulated by the generic kernels while the
    plan-specialised kernel is built on a background thread; the next sweep of the same shape loads it from the cache.
    Identical results either way."""
    from asyncflow_amd import build as af_build
    from asyncflow_amd import jit

    if not af_build.have_hipcc():
        pytest.skip("no hipcc on this box: nothing to build in the background")
    monkeypatch.setattr(jit, "CACHE_DIR", tmp_path)
    monkeypatch.setattr(jit, "_background", {})
    monkeypatch.setenv("ASYNCFLOW_JIT_BACKGROUND", "1")
    payload = lb_two_servers(horizon=45)
    seeds = np.arange(64, dtype=np.uint64) + 4242
    first = _runner(payload, seeds=seeds).run()
    assert first.engine_stats.flow_scenarios == 64 and first.engine_stats.specialised_launches == 0
    assert len(jit._background) == 1                                  # noqa: SLF001
    for t in jit._background.values():                                # noqa: SLF001
        t.join(timeout=180)
        assert not t.is_alive()
    second = _runner(payload, seeds=seeds).run()
    assert second.engine_stats.specialised_launches >= 1 and second.engine_stats.jit_fallbacks == 0
    _same_batches(first, second)


@pytest.mark.parametrize("kw", [dict(front=1), dict(front=2, backend=True, spike=True), dict(front=1, algo="least_connection"),
                                dict(front=2, general=True, backend=True)],
                         ids=["gateway", "two-gateways-backend-events", "least-connections", "general-backend"])
def test_servers_in_front_of_the_load_balancer_run_on_the_flow_kernel(kw):
    """Round 5: client -> server chain -> LB -> servers [-> backend] -> client on the stage-parallel kernel (the
    LB station behind the levels in front of it), generic and plan-specialised builds (`AF_FJ_LB_POS`), against the next-event
    kernels and the oracle."""
    from oracle.scenarios import gateway_lb

    payload = gateway_lb(users=150, horizon=30, **kw)
    seeds = np.arange(32, dtype=np.uint64) + 77
    res = _runner(payload, seeds=seeds, specialise=False).run()
    assert res.flow_reason == "" and res.engine_stats.flow_scenarios == 32 and res.engine_stats.flow_to_next_event <= 2, res.flow_reason
    plan = lower(payload)
    for i in (0, 31):
        _assert_scenario(res[i], ol.simulate(plan, int(seeds[i])), f"{kw} scenario {i}")
    _same_batches(res, _runner(payload, seeds=seeds, flow=False).run())
    special = _runner(payload, seeds=seeds, specialise=True).run()
    assert special.engine_stats.specialised_launches >= 1
    _same_batches(res, special)


# ------------------------------------------------------------------------------------------------ sweeps in flight
def test_sweeps_in_flight_on_host_threads_equal_their_lone_runs():
    """Several sweeps at once on one GPU: every engine has its own HIP streams, counter block and buffers, the error string is
    per thread and ctypes releases the GIL inside `af_engine_run_summarized`, so runners called from host threads overlap on
    the device (the next sweep's pre-generation and first waves beside the previous sweep's last residency round:
    scripts/gpu_two_sweeps_in_flight.py measures it).  Three different payloads -- stage-parallel kernel with the analyzer in
    the same call, general servers, next-event kernels only -- each twice in flight, equal their lone runs and the oracle."""
    import threading

    cases = [(lb_two_servers(horizon=60), dict(seeds=np.arange(700, dtype=np.uint64) + 11, summary=True)),
             (fanout8(horizon=30), dict(seeds=np.arange(300, dtype=np.uint64) + 5)),
             (single_server(horizon=120), dict(seeds=np.arange(500, dtype=np.uint64) + 3, flow=False))]
    lone = [_runner(p, **kw).run() for p, kw in cases]
    got: dict[int, object] = {}
    errors: list[BaseException] = []

    def work(slot: int) -> None:
        try:
            p, kw = cases[slot % len(cases)]
            got[slot] = _runner(copy.deepcopy(p), **kw).run()
        except BaseException as exc:  # noqa: BLE001
            errors.append(exc)

    threads = [threading.Thread(target=work, args=(s,)) for s in range(2 * len(cases))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for slot in range(2 * len(cases)):
        want = lone[slot % len(cases)]
        _same_batches(got[slot], want)
        if cases[slot % len(cases)][1].get("summary"):
            a, b = got[slot].summary()["stats"].cpu().numpy(), want.summary()["stats"].cpu().numpy()
            assert np.array_equal(a.view(np.uint64), b.view(np.uint64))      # (the analyzer of the same call: bit patterns)
    for k, (p, kw) in enumerate(cases):
        plan = lower(p)
        for i in (0, len(kw["seeds"]) - 1):
            _assert_scenario(got[k][i], ol.simulate(plan, int(kw["seeds"][i])), f"case {k} scenario {i}")
