"""Host logic of the sweep front-end (no GPU)."""

from __future__ import annotations

import numpy as np
import pytest

from asyncflow_amd.plan import lower
from asyncflow_amd.runner import resolve_sweep
from asyncflow_amd.sweep import expand_grid
from oracle.scenarios import lb_two_servers

USERS = "rqs_input.avg_active_users.mean"
RTT = "topology_graph.edges[*].latency.mean"


def test_grid_is_the_cartesian_product_times_replicas():
    sw = expand_grid({USERS: [10, 20, 30], RTT: [0.001, 0.002]}, replicas=4, seed_base=100)
    assert len(sw) == 24 and sw.shape == (3, 2)
    assert sw.columns[USERS][:8].tolist() == [10.0] * 8 and sw.columns[RTT][:8].tolist() == [0.001] * 4 + [0.002] * 4
    assert sw.seeds.dtype == np.uint64 and sw.seeds.tolist() == list(range(100, 124))
    assert sw.point.tolist() == [i // 4 for i in range(24)] and sw.replica.tolist() == [0, 1, 2, 3] * 6
    back = sw.by_point(np.arange(24))
    assert back.shape == (3, 2, 4) and back[2, 1].tolist() == [20, 21, 22, 23]


def test_ordering_by_load_keeps_seeds_attached_to_their_point():
    plain = expand_grid({USERS: [10, 300, 50], RTT: [0.001, 0.002]}, replicas=2)
    heavy = expand_grid({USERS: [10, 300, 50], RTT: [0.001, 0.002]}, replicas=2, order_by_load=USERS)
    assert heavy.columns[USERS].tolist() == sorted(plain.columns[USERS].tolist(), reverse=True)
    key = lambda s: sorted(zip(s.seeds.tolist(), s.columns[USERS].tolist(), s.columns[RTT].tolist()))  # noqa: E731
    assert key(plain) == key(heavy)
    vals = heavy.columns[USERS] * 1000 + heavy.replica
    assert np.array_equal(heavy.by_point(vals), plain.by_point(plain.columns[USERS] * 1000 + plain.replica))


def test_columns_resolve_against_a_plan_and_are_validated():
    plan = lower(lb_two_servers(horizon=10))
    sw = expand_grid({USERS: [10, 20], RTT: [0.001]}, replicas=3)
    cols = resolve_sweep(plan, sw.columns, len(sw))
    assert len(cols) == 1 + plan.n_edges and all(c[2].shape == (6,) for c in cols)
    with pytest.raises(ValueError, match="positive"):
        resolve_sweep(plan, {RTT: [0.0] * 6}, 6)
    with pytest.raises(ValueError, match="positive"):
        resolve_sweep(plan, {"topology_graph.nodes.servers[srv-1].endpoints[0].steps[0].cpu_time": [-1.0] * 6}, 6)
    with pytest.raises(ValueError, match="unsupported"):
        resolve_sweep(plan, {"sim_settings.total_simulation_time": [5.0] * 6}, 6)
    with pytest.raises(ValueError):
        expand_grid({USERS: []})
    with pytest.raises(ValueError):
        expand_grid({USERS: [1.0]}, order_by_load=RTT)


# ------------------------------------------------------------------------------------------------
# round 3: event parameters, server resources and the sampling window as sweep axes
# ------------------------------------------------------------------------------------------------
import copy  # noqa: E402

from asyncflow_amd.runner import write_point  # noqa: E402
from asyncflow_amd.workloads import lb_with_events  # noqa: E402
from oracle import oracle_lib as ol  # noqa: E402
from tests.hostcheck import build as hc  # noqa: E402

SPIKE = "events[ev-spike-1].start.spike_s"
DOWN_T0 = "events[ev-srv1-down].start.t_start"
DOWN_T1 = "events[ev-srv1-down].end.t_end"
CORES = "topology_graph.nodes.servers[srv-1].server_resources.cpu_cores"
RAM = "topology_graph.nodes.servers[srv-2].server_resources.ram_mb"
WINDOW = "rqs_input.user_sampling_window"


def _point_payload(plan, columns, i):
    """The payload a user of the reference would have written for sweep point i."""
    p = copy.deepcopy(plan.payload)
    for key, col in columns.items():
        write_point(p, key, np.broadcast_to(np.asarray(col, dtype=np.float64), (len(col),))[i] if np.ndim(col) else col)
    return p


def _engine_overrides(cols, i):
    names = {v: k for k, v in __import__("asyncflow_amd")._abi.PARAM_CODES.items()}
    return [(names[c], idx, float(v[i])) for c, idx, v, _ in cols]


def test_event_resource_and_window_axes_become_engine_columns():
    plan = lower(lb_with_events(users=200, horizon=60, scale=0.1))
    n = 6
    sweep = {SPIKE: np.linspace(0.01, 0.06, n), CORES: [1, 2, 3, 1, 2, 3], RAM: [1024, 2048] * 3, WINDOW: 30}
    cols = resolve_sweep(plan, sweep, n)
    by_code = {}
    for c, idx, v, _ in cols:
        by_code.setdefault(c, []).append((idx, v))
    code = __import__("asyncflow_amd")._abi.PARAM_CODES
    assert [i for i, _ in by_code[code["srv_cores"]]] == [0] and [i for i, _ in by_code[code["srv_ram_mb"]]] == [1]
    assert by_code[code["gen_window"]][0][1].tolist() == [30.0] * n
    # the spike of ev-spike-1 sits in two mark slots (start: +spike, end: -spike); nothing else changes
    deltas = sorted(by_code[code["emark_delta"]])
    assert len(deltas) == 2 and np.allclose(deltas[0][1], -deltas[1][1]) and code["emark_time"] not in by_code
    assert code["smark_time"] not in by_code


def test_an_outage_window_that_overtakes_another_event_reorders_the_timeline_slots():
    """srv-1's outage is swept from before to after the second spike's start: the server timeline keeps its slots,
    but WHICH event a slot belongs to changes with the scenario -- every slot gets its own columns."""
    base = lb_with_events(users=200, horizon=60, scale=0.1)
    plan = lower(base)
    t0 = np.array([18.0, 43.0, 50.0])     # (srv-2 is down during [36, 42]: both down at once is invalid)
    sweep = {DOWN_T0: t0, DOWN_T1: t0 + 3.0}
    cols = resolve_sweep(plan, sweep, 3)
    code = __import__("asyncflow_amd")._abi.PARAM_CODES
    edges = [v for c, _, v, _ in cols if c == code["smark_lb_edge"]]
    assert edges, "the slots change owner: their LB-edge columns must be present"
    for i in range(3):
        want = lower(_point_payload(plan, sweep, i))
        got = copy.deepcopy(plan)
        ol.apply_overrides(got, {(k, idx): v for k, idx, v in _engine_overrides(cols, i)})
        assert np.array_equal(got.smark_time, want.smark_time) and np.array_equal(got.smark_lb_edge, want.smark_lb_edge)
        assert np.array_equal(got.smark_down, want.smark_down)


def test_invalid_points_are_rejected_by_the_payload_models():
    plan = lower(lb_with_events(users=200, horizon=60, scale=0.1))
    with pytest.raises(ValueError, match="not a valid payload"):
        resolve_sweep(plan, {DOWN_T0: [18.0, 30.0], DOWN_T1: [24.0, 29.0]}, 2)          # t_start >= t_end (injection.py:94-98)
    with pytest.raises(ValueError, match="not a valid payload"):
        resolve_sweep(plan, {RAM: [2048, 128]}, 2)                                      # ram_mb >= 256 (nodes.py:65-68)
    with pytest.raises(ValueError, match="integer"):
        resolve_sweep(plan, {CORES: [1.5, 2]}, 2)
    with pytest.raises(ValueError, match="not a valid payload"):
        resolve_sweep(plan, {WINDOW: [30, 500]}, 2)                                     # window within [1, 120] s
    with pytest.raises(ValueError, match="not a network spike"):
        resolve_sweep(plan, {"events[ev-srv1-down].start.spike_s": [0.1, 0.2]}, 2)
    with pytest.raises(ValueError, match="unknown event"):
        resolve_sweep(plan, {"events[nope].start.t_start": [1.0, 2.0]}, 2)
    # both servers down at once leaves the load balancer without a target (payload.py / plan.lower)
    with pytest.raises(ValueError, match="not a valid payload"):
        resolve_sweep(plan, {DOWN_T0: [18.0, 36.5], DOWN_T1: [24.0, 41.0]}, 2)


def test_interior_points_are_validated_too():
    """The reference validates every payload it runs (schemas/payload.py:20-252): so does a sweep of up to VALIDATE_EVERY_POINT_UP_TO distinct
    points -- an invalid point between two valid extremes does not slip through (a fractional window)."""
    from asyncflow_amd import runner

    plan = lower(lb_with_events(users=200, horizon=60, scale=0.1))
    with pytest.raises(ValueError, match="integer field"):
        resolve_sweep(plan, {WINDOW: [60, 90.5, 120]}, 3)                              # an int field (rqs_generator.py:17-27)
    # both outages overlap only at the MIDDLE point: the extremes are valid payloads
    with pytest.raises(ValueError, match="not a valid payload"):
        resolve_sweep(plan, {DOWN_T0: [18.0, 36.5, 20.0], DOWN_T1: [24.0, 41.0, 25.0]}, 3)
    cols = {"rqs_input.avg_active_users.mean": np.repeat(np.arange(1.0, 41.0), 5)}    # 40 distinct points x 5 seeds
    assert runner.validate_points(plan, {k: np.asarray(v) for k, v in cols.items()}, 200) == 40
    # beyond VALIDATE_EVERY_POINT_UP_TO distinct points: every distinct VALUE of every column inside the first row
    # that holds it, then the rows holding the extremes, the ends and 64 rows spread evenly -- a 30 x 30 grid is 60 + < 70 rows, not 900
    a, b = np.meshgrid(np.linspace(10.0, 500.0, 30), np.linspace(0.001, 0.02, 30))
    grid = {"rqs_input.avg_active_users.mean": a.ravel(), "topology_graph.edges[*].latency.mean": b.ravel()}
    assert 900 > runner.VALIDATE_EVERY_POINT_UP_TO
    done = runner.validate_points(plan, grid, 900)
    assert 59 + 4 <= done <= 60 + 70
    assert runner.validate_points(plan, grid, 900) == done                                # (memoised: same plan, same columns)
    bad = {k: v.copy() for k, v in grid.items()}
    bad["topology_graph.edges[*].latency.mean"][437] = -0.001                             # one interior value of one column
    with pytest.raises(ValueError, match="not a valid payload|must be positive"):
        resolve_sweep(plan, bad, 900)
    with pytest.raises(ValueError, match="integer field"):                               # ... and integrality, whole column
        resolve_sweep(plan, {WINDOW: np.where(np.arange(10_002) == 5_000, 60.5, 60.0)}, 10_002)
    # columns that share a cross-field constraint.  1 000 points over a spike's start AND end, t_end = t_start + 5 in
    # every row: each row is a valid payload, although most t_start values lie behind the BASE payload's t_end -- a distinct value
    # is validated inside a row that holds it, never alone beside the base payload's other fields
    ev = {e["event_id"]: e for e in plan.payload["events"]}
    t0 = ev["ev-spike-1"]["start"]["t_start"] + np.linspace(0.0, 20.0, 1_000)
    moving = {"events[ev-spike-1].start.t_start": t0, "events[ev-spike-1].end.t_end": t0 + 5.0}
    assert t0[-1] > ev["ev-spike-1"]["end"]["t_end"] and 1_000 > runner.VALIDATE_EVERY_POINT_UP_TO
    assert runner.validate_points(plan, moving, 1_000) >= 1_000          # (every value is distinct here: every row is checked)
    crossed = {k: v.copy() for k, v in moving.items()}
    crossed["events[ev-spike-1].end.t_end"][613] = t0[613] - 0.5         # ONE interior row ends before it starts
    with pytest.raises(ValueError, match="not a valid payload"):
        runner.validate_points(plan, crossed, 1_000)


@pytest.mark.parametrize("kernel", ["next-event", "flow"])
def test_swept_points_match_the_oracle_run_on_the_written_out_payload(kernel):
    """Every new axis at once, on both kernel families (host builds: the next-event core for one lane, the
    stage-parallel kernel on the wave emulator): scenario i of the sweep == the oracle on the payload with the values
    of point i written into it -- the reference's own way of running that point."""
    base = lb_with_events(users=150, horizon=30, scale=0.05)
    base["topology_graph"]["nodes"]["servers"][0]["server_resources"]["cpu_cores"] = 2
    plan = lower(base)
    ev = {e["event_id"]: e for e in plan.payload["events"]}
    s1 = ev["ev-spike-1"]["start"]["t_start"]
    n = 5
    sweep = {
        SPIKE: np.array([0.004, 0.02, 0.05, 0.03, 0.01]),
        "events[ev-spike-1].start.t_start": s1 + np.array([0.0, 0.2, -0.3, 0.1, 0.0]),
        # (srv-2 is down during [18, 21]: point 2 moves srv-1's outage BEHIND it -- the timeline slots change owner)
        DOWN_T0: ev["ev-srv1-down"]["start"]["t_start"] + np.array([0.0, 0.5, 12.5, -1.0, 2.0]),
        DOWN_T1: ev["ev-srv1-down"]["end"]["t_end"] + np.array([0.0, 0.5, 13.0, 0.0, 3.0]),
        CORES: [1, 2, 3, 2, 1],
        RAM: [512, 2048, 1024, 640, 4096],
        WINDOW: [1, 7, 30, 60, 2],
    }
    cols = resolve_sweep(plan, sweep, n)
    for i in range(n):
        want = ol.simulate(lower(_point_payload(plan, sweep, i)), 1000 + i)
        ov = _engine_overrides(cols, i)
        if kernel == "flow":
            res = hc.flow_simulate(plan, 1000 + i, overrides=ov, ring_rows=256)
            assert res is not None, hc.flow_reason()
            counts, clock, samples = res
            if int(counts[5]) & hc.FLOW_FALLBACK:       # handed back (a tie): the next-event kernels take it on the GPU
                continue
        else:
            counts, clock, samples = hc.simulate(plan, 1000 + i, overrides=ov)
        assert np.array_equal(counts[:5].astype(np.uint64), want.counts[:5]), (i, counts, want.counts)
        assert int(counts[7]) == int(want.counts[7]), "injection marks applied"
        assert np.array_equal(clock.view(np.uint64), want.clock.view(np.uint64)), f"point {i}: rqs_clock"
        assert np.array_equal(samples, want.samples), f"point {i}: sampled series"
