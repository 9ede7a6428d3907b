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
