"""CPU differential tests of the STAGE-PARALLEL kernel (asyncflow_amd/csrc/af_flow.hpp).

The kernel moves 64 requests per step through the stations of the request path (one wave per
scenario).  tests/hostcheck/ runs the very same source on a 64-fibre wave emulator
(tests/hostcheck/wave_emul.hpp, test-only) so that its logic is checked here, without a GPU, against
the SimPy-faithful oracle (oracle/des_oracle.c, itself pinned on the reference's fixtures):

* whenever the kernel does NOT hand a scenario back (FLAG_FLOW_FALLBACK clear) its outputs are the
  oracle's bit for bit -- counts, every (start, finish) pair in completion order, every sample;
* what it cannot express (two events of one station at one instant, RAM admission that would block,
  list / tick-ring overflow) is handed back, never silently wrong.
"""

from __future__ import annotations

import copy
import json
import random

import numpy as np
import pytest

from asyncflow_amd import _abi
from asyncflow_amd.plan import lower
from asyncflow_amd.workloads import fanout8, lb_two_servers, lb_with_events, single_server, single_server_with_spike
from oracle import oracle_lib as ol
from oracle.scenarios import deep_chain, flow_payload, gateway_lb, server_chain, server_tiers, stress_mixed, wide_fanout
from tests.conftest import GOLDEN_DIR
from tests.hostcheck import build as hc


def _run(payload, seed, **kw):
    plan = lower(payload)
    got = hc.flow_simulate(plan, seed, **kw)
    assert got is not None, hc.flow_reason()
    counts, clock, samples = got
    flags = int(counts[_abi.CNT_FLAGS])
    if flags & hc.FLOW_FALLBACK:
        return "fallback", {v for k, v in hc.FLOW_WHY.items() if flags & k}
    want = ol.simulate(plan, seed)
    assert np.array_equal(want.counts[:5].astype(np.uint32), counts[:5]), (want.counts[:8], counts[:8])
    assert int(counts[_abi.CNT_MARKS]) == int(want.counts[_abi.CNT_MARKS])
    assert np.array_equal(want.clock.view(np.uint64), clock.view(np.uint64))
    assert np.array_equal(want.samples, samples)
    assert (flags & _abi.FATAL_FLAGS) == 0
    assert (flags & 0xFF) == (int(want.counts[_abi.CNT_FLAGS]) & 0xFF)      # informational flags too (RAM_STARVED)
    return "exact", want


@pytest.mark.parametrize("name", ["single_server_t30", "lb2_rr_t30", "lb2_events_t60", "fanout8_t20"])
def test_flow_kernel_reproduces_the_reference_fixtures(name):
    fx = np.load(GOLDEN_DIR / f"{name}.npz", allow_pickle=False)
    payload = json.loads(str(fx["payload_json"]))
    plan = lower(payload)
    counts, clock, samples = hc.flow_simulate(plan, int(fx["seed"]), ipl=2, ring_rows=256)
    assert not int(counts[_abi.CNT_FLAGS]) & hc.FLOW_FALLBACK
    assert (int(counts[0]), int(counts[1]), int(counts[2]), int(counts[4])) == \
        (int(fx["generated"]), int(fx["completed"]), int(fx["dropped"]), int(fx["ticks"]))
    assert np.array_equal(clock, fx["clock"]) and np.array_equal(samples, fx["samples"])   # the reference's own output


@pytest.mark.parametrize(("ring_rows", "far"), [(64, True), (64, False), (0, True)])
def test_baseline_workloads_bit_exact(ring_rows, far):
    """tick differences in the LDS ring (lean instantiations with / without FEAT_FAR) / in the sample rows (HBM)"""
    assert _run(lb_two_servers(horizon=60), 0x5EED0000, ring_rows=ring_rows, far=far)[0] == "exact"
    assert _run(single_server(horizon=60), 0, ring_rows=ring_rows, far=far)[0] == "exact"
    assert _run(lb_with_events(users=300, horizon=60, scale=0.1), 42, ring_rows=ring_rows, far=far)[0] == "exact"
    assert _run(fanout8(horizon=40), 11, ipl=2, ring_rows=256 if ring_rows else 0, far=far)[0] == "exact"
    if ring_rows and far:      # (the ring a sweep of this plan gets on the device: 32 rows = 1.6 s for ~4-s requests)
        assert _run(fanout8(horizon=40), 12, ipl=2, ring_rows=32)[0] == "exact"


def test_grid_corners_of_configs_3_and_4():
    """users 10 / 1000 x per-hop latency 0.5 / 50 ms, with and without the injected events."""
    for users, hop in ((10, 0.0005), (1000, 0.0005), (10, 0.05), (1000, 0.05)):
        for base in (lb_two_servers(horizon=20), lb_with_events(users=400, horizon=20, scale=20 / 600)):
            p = copy.deepcopy(base)
            p["rqs_input"]["avg_active_users"]["mean"] = users
            for e in p["topology_graph"]["edges"]:
                e["latency"]["mean"] = hop
            status, _ = _run(p, 0xC0F30000 + users, ipl=2, ring_rows=128)
            assert status == "exact", (users, hop)


def test_per_scenario_overrides_reach_the_kernel():
    """af_override_t columns (users, an edge's mean and dropout, a step time) against the oracle run on
    the payload that has those values written into it."""
    base = lb_two_servers(horizon=20)
    plan = lower(base)
    ov = [("gen_users_mean", 0, 150.0), ("edge_mean", 1, 0.01), ("edge_dropout", 2, 0.2), ("step_time", 0, 0.004)]
    counts, clock, samples = hc.flow_simulate(plan, 9, overrides=ov)
    edited = copy.deepcopy(base)
    edited["rqs_input"]["avg_active_users"]["mean"] = 150.0
    edited["topology_graph"]["edges"][1]["latency"]["mean"] = 0.01
    edited["topology_graph"]["edges"][2]["dropout_rate"] = 0.2
    edited["topology_graph"]["nodes"]["servers"][0]["endpoints"][0]["steps"][0]["step_operation"] = {"cpu_time": 0.004}
    want = ol.simulate(lower(edited), 9)
    assert not int(counts[_abi.CNT_FLAGS]) & hc.FLOW_FALLBACK
    assert np.array_equal(want.counts[:5].astype(np.uint32), counts[:5])
    assert np.array_equal(want.clock, clock) and np.array_equal(want.samples, samples)


def test_small_lists_hand_the_scenario_back_instead_of_dropping_messages():
    status, why = _run(fanout8(horizon=30), 11, ipl=1, ring_rows=256)      # ~40 messages in flight per edge
    assert status == "fallback" and "list" in why
    # FEAT_FAR: a delivery the ring does not reach is entered by the receiving station when it handles it, so 1-s hops
    # do not need a ring that reaches 1 s ahead (0.4 s of ring here); without it the sender enters both ends: handed back
    assert _run(fanout8(horizon=30), 11, ipl=2, ring_rows=8)[0] == "exact"
    status, why = _run(fanout8(horizon=30), 11, ipl=2, ring_rows=8, far=False)
    assert status == "fallback" and "ring" in why
    # ... only the time a request spends INSIDE a server has to fit: a saturated server (1 400 rps on 1 000 rps of core)
    status, why = _run(single_server(users=700, rpm=120, horizon=6), 3, ipl=4, ring_rows=8)
    assert status == "fallback" and "ring" in why


def test_plans_outside_the_feed_forward_range_are_refused():
    odd_ram = lb_two_servers(horizon=10)
    odd_ram["topology_graph"]["nodes"]["servers"][0]["endpoints"][0]["steps"][1]["step_operation"]["necessary_ram"] = 100.1
    # (servers feeding servers are in range since round 4, a server chain in front of the LB since round 5; the client AND a server
    # feeding the LB is not)
    feeds_lb = gateway_lb(front=1, horizon=10)
    feeds_lb["topology_graph"]["edges"][1]["target"] = "lb"
    for payload, word in ((odd_ram, "1/256 MB"), (wide_fanout(horizon=12), "16 servers"), (feeds_lb, "both feed the load balancer")):
        assert hc.flow_simulate(lower(payload), 1) is None
        assert word in hc.flow_reason()


# ------------------------------------------------------------------------------------------------ servers that feed servers
def test_server_tiers_run_level_by_level():
    """FEAT_CHAIN (round 3): client -> s0 -> s1 -> client.  The servers are put in levels and the server station runs once
    per level and round over the ONE server list, each level with its own horizon: everything a level-k server receives
    before the horizon of level k - 1 is in the list, because a request leaves a server no earlier than it arrived.
    Every instantiation the engine would launch (64- / 128-entry lists, LDS ring / differences in HBM, the second-chance
    form) against the oracle: counts, every (start, finish) pair, every sample."""
    p = server_chain("exponential", 0.003, cores=2, horizon=20)
    for s in p["topology_graph"]["nodes"]["servers"]:                     # (continuous step times: server_chain's are dyadic tie makers)
        for st in s["endpoints"][0]["steps"]:
            op = st["step_operation"]
            for k in ("cpu_time", "io_waiting_time"):
                if k in op:
                    op[k] = op[k] * 0.013
    p["rqs_input"]["avg_active_users"]["mean"] = 80
    for kw in (dict(ipl=1, ring_rows=64), dict(ipl=2, ring_rows=0), dict(robust=True, ring_rows=0)):
        status, want = _run(p, 7, **kw)
        assert status == "exact" and want.completed > 1000, (kw, status)


@pytest.mark.parametrize("block", range(3))
def test_fuzzed_server_tiers_are_exact_or_handed_back(block):
    """oracle/scenarios.py::server_tiers: [LB ->] front servers -> [middle ->] backend -> client in up to three levels, server
    indices in any order, a front server may answer the client itself, the LB may feed the backend too, four latency
    laws, RAM pressure, dyadic RAM needs, a spike on any edge, outages.  The lean form may hand back (second-long
    log-normal hops outgrow 128-entry lists); the second-chance form with long lists must not."""
    exact = 0
    for case in range(block * 20, block * 20 + 20):
        rng = random.Random(91000 + case)
        p = server_tiers(rng)
        status, _ = _run(p, 300 + case, ipl=2, ring_rows=64)
        assert status in ("exact", "fallback")
        exact += status == "exact"
        assert _run(p, 300 + case, robust=True, ring_rows=0, long_list_entries=1024)[0] == "exact", case
    assert exact >= 14


def test_server_tiers_behind_a_least_connections_lb():
    """Round 4: the least-connections walk counts the entries of the server list that came by the LB's OWN edges (a list
    entry carries its in-edge): what the front servers send to the backend is in the same list and is not in flight on an
    LB edge (lb_algorithms.py:10-20 counts edge.concurrent_connections of the LB's out-edges)."""
    exact = with_lb = 0
    for case in range(24):
        p = server_tiers(random.Random(94000 + case), algo="least_connection")
        if "load_balancer" not in p["topology_graph"]["nodes"]:
            continue
        with_lb += 1
        status, _ = _run(p, 700 + case, ipl=2, ring_rows=64)
        assert status in ("exact", "fallback")
        exact += status == "exact"
        assert _run(p, 700 + case, robust=True, ring_rows=0, long_list_entries=1024)[0] == "exact", case
    assert with_lb >= 8 and exact >= with_lb - 4
    # ... and with general servers in the tiers (FEAT_GENSRV | FEAT_LC | FEAT_CHAIN)
    for case in range(24, 40):
        p = server_tiers(random.Random(94000 + case), general=True, algo="least_connection")
        assert _run(p, 700 + case, ipl=1, ring_rows=32)[0] in ("exact", "fallback")
        assert _run(p, 700 + case, robust=True, ring_rows=0, long_list_entries=1024)[0] == "exact", case


@pytest.mark.parametrize("block", range(2))
def test_fuzzed_tiers_of_general_servers_are_exact_or_handed_back(block):
    """Round 4: server tiers whose servers have two endpoints or come back to the core queue (FEAT_GENSRV | FEAT_CHAIN): the
    event-by-event station runs the servers of each level up to that level's horizon, and what they send goes to the
    completion list or back into the server list.  The compact first-launch form and the second-chance form."""
    exact = 0
    for case in range(block * 15, block * 15 + 15):
        p = server_tiers(random.Random(93000 + case), general=True)
        status, _ = _run(p, 500 + case, ipl=1, ring_rows=32)
        assert status in ("exact", "fallback")
        exact += status == "exact"
        assert _run(p, 500 + case, robust=True, ring_rows=0, long_list_entries=1024)[0] == "exact", case
    assert exact >= 12


@pytest.mark.parametrize("block", range(6))
def test_fuzzed_feed_forward_payloads_are_exact_or_handed_back(block):
    """Idle to saturated, multi-core, leading / trailing I/O, dyadic step times, tight RAM, spikes, outages."""
    exact = 0
    for case in range(block * 25, block * 25 + 25):
        rng = random.Random(31000 + case)
        status, _ = _run(flow_payload(rng, horizon=6), 900 + case, ipl=4, ring_rows=1024)
        exact += status == "exact"
    assert exact >= 4, f"only {exact} of 25 fuzzed payloads ran on the flow kernel"


@pytest.fixture
def quantised_times():
    """TEST-ONLY switch of the oracle and of the host builds: latencies and arrival gaps rounded down to a multiple
    of 2**-bits s, so that exact timestamp ties happen by the thousand (with continuous laws: ~1e-6 per scenario)."""
    def switch(bits: int) -> None:
        hc.set_test_quantum(bits)
        ol.set_test_quantum(bits)
    yield switch
    switch(0)


@pytest.mark.parametrize("bits", [12, 16])
def test_exact_ties_by_the_thousand_follow_simpy_order(quantised_times, bits):
    """Equal delivery times at a station, deliveries at their own send instant, arrivals on core / RAM releases, events on
    ticks: the lean instantiation hands such scenarios back; the second-chance instantiation (messages carry their send
    time) orders two deliveries of one station the way SimPy pops them -- by creation of their Timeouts = by send time --
    and must then equal the SimPy-faithful oracle bit for bit.  Never a silent difference in either."""
    quantised_times(bits)
    cases = [(lb_two_servers(horizon=15), s) for s in range(4)] + [(lb_with_events(users=300, horizon=30, scale=0.05), 7)]
    cases += [(single_server(horizon=20), 3), (fanout8(horizon=15), 5)]
    cases += [(flow_payload(random.Random(31000 + k), horizon=5), 900 + k) for k in range(24)]
    ties = resolved = 0
    for payload, seed in cases:
        plan = lower(payload)
        want = ol.simulate(plan, seed, clock_capacity=4 * plan.clock_capacity())
        if int(want.counts[_abi.CNT_FLAGS]) & _abi.FATAL_FLAGS:
            continue
        ties += int(ol.lib().orc_last_ties())
        for kw in (dict(ipl=2, ring_rows=256), dict(robust=True, ring_rows=0)):
            counts, clock, samples = hc.flow_simulate(plan, seed, clock_capacity=4 * plan.clock_capacity(),
                                                      draw_capacity=4 * plan.clock_capacity(), **kw)
            if int(counts[_abi.CNT_FLAGS]) & (hc.FLOW_FALLBACK | _abi.FATAL_FLAGS):
                continue
            assert np.array_equal(want.counts[:5].astype(np.uint32), counts[:5]), (seed, kw)
            assert np.array_equal(want.clock.view(np.uint64), clock.view(np.uint64)), (seed, kw)
            assert np.array_equal(want.samples, samples), (seed, kw)
            resolved += bool(kw.get("robust"))
    assert ties > 2_000 and resolved >= 5, (ties, resolved)


@pytest.mark.parametrize("heavy", [False, True])
def test_second_long_spikes_of_the_reference_examples(heavy):
    """event_inj_single_server.yml / heavy_inj_single_server.yml: a 2 s / 3 s spike on the client -> server edge.

    While the spike lasts the server station runs AHEAD of the client by the spike (Flow::send_floor); when it ends,
    rate x spike messages wait at the servers and then in the completion list: register-resident lists hand that
    back, the long-list instantiation (FEAT_BIGLIST, per-list capacities) carries it, bit for bit.
    The heavy file is RAM-bound with one core and fixed step times: a request is admitted at the very instant its
    predecessor frees a core (G[j-slots] == F[j-1], the same f64 sum) all the time -- which must not hand back."""
    p = single_server_with_spike(heavy=heavy, horizon=60, scale=0.1)     # spike from t = 12 / 18 s to 24 / 30 s
    seed = 0x5EED0002
    st, why = _run(p, seed, ring_rows=0)
    if heavy:
        assert (st, why) == ("fallback", {"list"})                          # 450 messages do not fit 64 entries
    for kw in (dict(long_list_entries=1024), dict(long_list_entries=1024, long_list=2)):
        st, got = _run(p, seed, ring_rows=0, robust=True, **kw)
        if "long_list" in kw and heavy:
            assert (st, got) == ("fallback", {"list"})                      # the completion list needs the room too
        else:
            assert st == "exact"
            assert got.completed > (2500 if heavy else 500)


def test_long_lists_on_both_sides_of_the_register_ranked_length():
    """select_big() ranks a list of up to 128 entries out of registers (round 6) and walks a longer one in chunks: hops of
    0.3 / 0.9 / 2 s at 133 requests per second keep ~40 / ~120 / ~270 messages pending per station, so the lists cross that
    length in both directions inside one run."""
    for hop, seed in ((0.3, 5), (0.9, 6), (2.0, 7)):
        p = lb_two_servers(horizon=30)
        for e in p["topology_graph"]["edges"]:
            e["latency"]["mean"] = hop
        assert _run(p, seed, ipl=1, ring_rows=0, robust=True, long_list_entries=1024)[0] == "exact", hop


def test_long_list_instantiation_matches_the_register_resident_one():
    """FEAT_BIGLIST changes how a list is walked, not what is selected: same outputs on workloads both can run."""
    for payload, seed in ((lb_two_servers(horizon=30), 7), (lb_with_events(users=300, horizon=60, scale=0.1), 42), (fanout8(horizon=20), 11)):
        assert _run(payload, seed, ring_rows=0, robust=True, long_list_entries=64)[0] in ("exact", "fallback")
        assert _run(payload, seed, ring_rows=0, robust=True, long_list_entries=512)[0] == "exact"


@pytest.mark.parametrize("kw", [dict(ipl=1, ring_rows=64), dict(ipl=2, ring_rows=0), dict(robust=True, ring_rows=0)])
def test_least_connections_is_a_decision_of_the_lb_station_alone(kw):
    """lb_algorithms.py:10-20 picks the out-edge with the fewest messages IN FLIGHT ON THE EDGE (edge.py:90,115):
    sent by the LB before t, delivered after t -- no feedback from the servers, so the plan stays feed-forward.
    Flow::lb_pick_lc walks the batch one message at a time against the server list; outages reorder the candidates."""
    assert _run(lb_two_servers(horizon=30, algo="least_connection"), 5, **kw)[0] == "exact"
    p = lb_with_events(users=300, horizon=60, scale=0.1)
    p["topology_graph"]["nodes"]["load_balancer"]["algorithms"] = "least_connection"
    assert _run(p, 42, **kw)[0] == "exact"
    p = fanout8(horizon=20)
    p["topology_graph"]["nodes"]["load_balancer"]["algorithms"] = "least_connection"
    assert _run(p, 11, **dict(kw, ring_rows=0, ipl=2))[0] == "exact"      # (~1-s hops: 128-entry lists, differences in HBM)


def test_poisson_integer_second_latencies_run_on_the_flow_kernel():
    """Round 3: Poisson edge latencies (samplers/common_helpers.py:67-70: whole seconds, zero included)
    are inside the stage-parallel kernel's range.  Fuzzed feed-forward payloads with 1-3 Poisson edges: wherever the
    kernel does not hand the scenario back (lists too short for rate x 1 s of waiting messages, a genuine tie) it
    equals the oracle bit for bit, and with long lists most of them stay."""
    import random

    from oracle.scenarios import flow_payload

    exact = back = 0
    for case in range(40):
        rng = random.Random(52000 + case)
        p = flow_payload(rng, horizon=8)
        edges = p["topology_graph"]["edges"]
        for e in rng.sample(edges, k=min(len(edges), rng.randint(1, 3))):
            e["latency"] = {"mean": rng.choice([0.05, 0.2, 0.5, 0.8]), "distribution": "poisson"}
        status, _ = _run(p, 900 + case, ipl=4, ring_rows=0, robust=True, long_list_entries=1024)
        exact += status == "exact"
        back += status == "fallback"
    assert exact >= 30 and exact + back == 40


@pytest.mark.parametrize("n_srv", [9, 12, 13, 16])
def test_round_robin_fan_out_beyond_eight_servers(n_srv):
    """Round 3: 9..12 servers behind a round-robin load balancer run on the stage-parallel kernel (16 slots
    per per-server array).  Round 5: 13..16 servers too -- 2 + 5 S sampled series are more than a wave has lanes, so a lane
    carries the running values of TWO series (Flow::flush_ticks, run_val2).  wide_fanout's topology with ONE endpoint per
    server: multi-core servers, RAM, outages and a spike."""
    from oracle.scenarios import wide_fanout

    payload = wide_fanout(n_srv, "round_robin", horizon=12, users=100)     # (odd servers answer over ~1-s log-normal hops)
    for s in payload["topology_graph"]["nodes"]["servers"]:
        s["endpoints"] = s["endpoints"][:1]
    assert _run(payload, 77, ipl=4, ring_rows=0)[0] == "exact"
    assert _run(payload, 78, ipl=4, ring_rows=128)[0] == "exact"
    if n_srv == 16:
        seventeen = wide_fanout(17, "round_robin", horizon=12)
        for s in seventeen["topology_graph"]["nodes"]["servers"]:
            s["endpoints"] = s["endpoints"][:1]
        assert hc.flow_simulate(lower(seventeen), 1) is None and "more than 16 servers" in hc.flow_reason()


@pytest.mark.parametrize("n_srv", [9, 12, 16])
def test_least_connections_fan_out_beyond_eight_servers(n_srv):
    """Round 6: a least-connections load balancer in front of 9 .. 16 servers runs on the stage-parallel kernel (the walk keeps
    16-bit in-flight counts of sixteen servers in four words and sixteen prepared draws per lane: Flow::lb_pick_lc_n<16>;
    lb_algorithms.py:10-20).  wide_fanout's topology with ONE endpoint per server: multi-core servers, RAM, two outages and a
    spike; then with both endpoints (general servers behind least connections)."""
    from oracle.scenarios import wide_fanout

    payload = wide_fanout(n_srv, "least_connection", horizon=10, users=100)
    both = copy.deepcopy(payload)
    for s in payload["topology_graph"]["nodes"]["servers"]:
        s["endpoints"] = s["endpoints"][:1]
    assert _run(payload, 81, ipl=4, ring_rows=0)[0] == "exact"
    assert _run(payload, 82, ipl=4, ring_rows=128)[0] == "exact"
    status, _ = _run(both, 83, ipl=1, ring_rows=0, robust=True, long_list_entries=1024)
    assert status in ("exact", "fallback")
    if n_srv == 16:
        seventeen = wide_fanout(17, "least_connection", horizon=10)
        assert hc.flow_simulate(lower(seventeen), 1) is None


@pytest.mark.parametrize(("depth", "fan"), [(4, True), (5, True), (5, False)])
def test_server_chains_of_four_and_five_levels(depth, fan):
    """Round 5: tiers deeper than three levels.  The server station runs once per level (FEAT_CHAIN), the
    levels' horizons are slots 4 .. 7; the generic instantiations keep them in LDS (Flow::hz) because a run-time choice among
    eight members put the whole object in scratch.  Six levels are refused (the next-event kernels run them)."""
    payload = deep_chain(depth, users=150, horizon=20, fan=fan)
    for seed in (3, 4):
        for kw in (dict(ipl=2, ring_rows=64), dict(ipl=1, ring_rows=0, robust=True, long_list_entries=512)):
            status, why = _run(payload, seed, **kw)
            assert status == "exact", (depth, fan, seed, kw, why)
    if depth == 5 and fan:
        assert hc.flow_simulate(lower(deep_chain(6, horizon=10)), 1) is None and "deeper than five levels" in hc.flow_reason()


@pytest.mark.parametrize("kw", [dict(front=1), dict(front=2, backend=True), dict(front=1, algo="least_connection", spike=True),
                                dict(front=1, general=True), dict(front=2, general=True, backend=True, spike=True)],
                         ids=["gateway", "two-gateways-backend", "least-connections-events", "general", "general-backend-events"])
def test_servers_in_front_of_the_load_balancer(kw):
    """Round 5 (`graph.py:135-157`): client -> server chain -> LB -> servers [-> backend] -> client.  The LB station
    runs BEHIND the server levels in front of it (Flow::lb_pos), the front servers' responses go into the LB's list, everything behind
    the LB is deeper than what feeds it; tandem and general servers, round robin and least connections, a spike on the edge into the LB
    and an outage behind it.  Exact in the lean and in the second-chance form; the client AND a server feeding the LB is refused."""
    payload = gateway_lb(users=150, horizon=15, **kw)
    for seed in (5, 6):
        for form in (dict(ipl=2, ring_rows=64), dict(ipl=1, ring_rows=0, robust=True, long_list_entries=512)):
            status, why = _run(payload, seed, **form)
            assert status == "exact", (kw, seed, form, why)
    if kw == dict(front=1):
        both = gateway_lb(front=1, horizon=10)
        both["topology_graph"]["edges"][1]["target"] = "lb"          # cli -> lb as well as gw0 -> lb
        assert hc.flow_simulate(lower(both), 1) is None and "both feed the load balancer" in hc.flow_reason()


def test_general_servers_several_endpoints_and_core_re_entry():
    """Round 3: several endpoints per server (server.py:101: one uniform draw per arriving request), step
    programs that come back to the core queue after an I/O step, RAM needs that differ per request (also in dyadic
    fractions of a MB) run on the stage-parallel kernel: its server station then simulates each server event by event
    (Flow::gen_servers, lane k = server k) up to the station's horizon.  The reference's own generality payloads and the
    fuzzed topologies of the next-event tests: exact, or handed back (two events of one server at one instant, more than
    32 requests inside one server, lists) -- never different."""
    import random

    from oracle.scenarios import overload, random_payload

    kw = dict(ipl=1, ring_rows=0, robust=True, long_list_entries=1024)
    assert _run(stress_mixed(40), 3, **kw)[0] in ("exact", "fallback")
    assert _run(overload(12), 5, **kw)[0] in ("exact", "fallback")
    two_ep = wide_fanout(8, "round_robin", horizon=12, users=100)         # two endpoints per server, multi-core, outages, a spike
    assert _run(two_ep, 1, **kw)[0] == "exact"
    exact = 0
    for case in range(40):
        status, _ = _run(random_payload(random.Random(31000 + case), horizon=8), 17 * case, **kw)
        exact += status == "exact"
    assert exact >= 25, f"only {exact} of 40 fuzzed topologies stayed on the stage-parallel kernel"


def test_general_servers_compact_first_launch():
    """Round 4: the FIRST launch of a plan with general servers uses the long-list instantiation WITHOUT send times and with
    lists sized by the load (128 entries here) -- 20 instead of 41 KB of LDS per wave, 7 instead of 3 waves per compute unit for a
    station that is one busy lane per server (engine.hip: plan_flow, gen_compact).  Exact or handed back to the second-chance
    form (256-entry lists with send times), never different; LDS tick ring or differences in HBM."""
    import random

    from asyncflow_amd.workloads import _endpoint
    from oracle.scenarios import random_payload, tie_storm

    report = lb_two_servers(horizon=40)
    for s in report["topology_graph"]["nodes"]["servers"]:
        s["endpoints"].append(_endpoint("/report", [("io_db", 0.004), ("ram", 64), ("cpu_bound_operation", 0.0015),
                                                     ("io_wait", 0.006), ("cpu_bound_operation", 0.0005)]))
    for ring in (0, 32):
        assert all(_run(report, 0x5EED0000 + seed, ipl=1, ring_rows=ring)[0] == "exact" for seed in range(4))
    exact = 0
    for case in range(16):
        exact += _run(random_payload(random.Random(31000 + case), horizon=8), 17 * case, ipl=1, ring_rows=0)[0] == "exact"
        exact += _run(tie_storm(random.Random(7000 + case), horizon=8), 3 * case + 1, ipl=1, ring_rows=32)[0] == "exact"
    assert exact >= 16, f"only {exact} of 32 fuzzed general-server payloads stayed on the compact form"


def test_general_servers_shared_instants_follow_simpy_order():
    """Round step times (1 ms CPU, 10 ms I/O, ...) make step ends of DIFFERENT requests of one server coincide all the
    time once requests queue for the core (their times are a common base + sums of step times).  Round 4: the general
    server station runs such an instant the way SimPy does -- every Timeout of the instant in creation order, then ONE FIFO
    of the zero-time steps they schedule (the Puts that give a core or the RAM back, the Gets that were granted):
    Flow::gs_instant, af_core.hpp's micro_mode restricted to one server -- so the order in which new Timeouts are created,
    waiters are served and responses leave is SimPy's, and later ties among THOSE are exact too.  What is still handed back:
    an arrival exactly at a step end (its place is decided by events of other nodes), > 32 requests inside one server.
    LB-2 with a second endpoint: round 3 handed back 5.6 % of the scenarios at T = 120 s and 42 % at 600 s; now none of
    40 / 24 seeds (the 600-s run is scripts-only: 5 s per seed on the emulator).  Tie storms (dyadic step times, Poisson
    latencies, tight RAM): exact or handed back, never different (storm 30 / seed 91 showed the RAM-admission order)."""
    from asyncflow_amd.workloads import _endpoint
    from oracle.scenarios import tie_storm

    report = lb_two_servers(horizon=40)
    for s in report["topology_graph"]["nodes"]["servers"]:
        s["endpoints"].append(_endpoint("/report", [("io_db", 0.004), ("ram", 64), ("cpu_bound_operation", 0.0015),
                                                     ("io_wait", 0.006), ("cpu_bound_operation", 0.0005)]))
    kw = dict(ipl=1, ring_rows=0, robust=True, long_list_entries=256)
    stayed = sum(_run(report, 0x5EED0000 + seed, **kw)[0] == "exact" for seed in range(8))
    assert stayed == 8, f"only {stayed} of 8 two-endpoint LB-2 scenarios stayed on the stage-parallel kernel"
    kw["long_list_entries"] = 1024
    assert _run(tie_storm(random.Random(7030), horizon=8), 91, **kw)[0] == "exact"
    ties = exact = 0
    for case in range(24):
        status, info = _run(tie_storm(random.Random(7000 + case), horizon=8), 3 * case + 1, **kw)
        exact += status == "exact"
        ties += status == "fallback" and "tie" in info
    assert exact >= 13 and ties <= 3, f"{exact} of 24 tie storms stayed on the stage-parallel kernel, {ties} handed back for a tie"
