"""Payload normalisation + lowering (host logic of the drop-in boundary)."""

from __future__ import annotations

import copy

import numpy as np
import pytest
import yaml

from asyncflow_amd import _abi
from asyncflow_amd.payload import normalize_payload
from asyncflow_amd.plan import estimate_capacities, lower
from asyncflow_amd.runner import resolve_sweep
from oracle.scenarios import lb_two_servers, lb_with_events, single_server, stress_mixed

LB2_YAML = """
rqs_input:
  id: rqs-1
  avg_active_users: { mean: 400 }
  avg_request_per_minute_per_user: { mean: 20 }
topology_graph:
  nodes:
    client: { id: client-1 }
    load_balancer: { id: lb-1, server_covered: [srv-1, srv-2] }
    servers:
      - id: srv-1
        server_resources: { cpu_cores: 1, ram_mb: 2048 }
        endpoints:
          - endpoint_name: /API
            steps:
              - { kind: initial_parsing, step_operation: { cpu_time: 0.002 } }
              - { kind: ram, step_operation: { necessary_ram: 128 } }
              - { kind: io_wait, step_operation: { io_waiting_time: 0.012 } }
      - id: srv-2
        server_resources: { cpu_cores: 1, ram_mb: 2048 }
        endpoints:
          - endpoint_name: /api
            steps:
              - { kind: initial_parsing, step_operation: { cpu_time: 0.002 } }
  edges:
    - { id: gen-client, source: rqs-1, target: client-1, latency: { mean: 0.003, distribution: exponential } }
    - { id: client-lb, source: client-1, target: lb-1, latency: { mean: 0.002, distribution: log_normal } }
    - { id: lb-srv1, source: lb-1, target: srv-1, latency: { mean: 0.002, distribution: exponential } }
    - { id: lb-srv2, source: lb-1, target: srv-2, latency: { mean: 0.002, distribution: exponential }, dropout_rate: 0 }
    - { id: srv1-client, source: srv-1, target: client-1, latency: { mean: 0.003, distribution: exponential } }
    - { id: srv2-client, source: srv-2, target: client-1, latency: { mean: 0.003, distribution: exponential } }
sim_settings:
  total_simulation_time: 600
"""


def test_defaults_follow_the_reference_schema():
    p = normalize_payload(yaml.safe_load(LB2_YAML))
    assert p["rqs_input"]["user_sampling_window"] == 60
    assert p["rqs_input"]["avg_active_users"]["distribution"] == "poisson"
    edges = {e["id"]: e for e in p["topology_graph"]["edges"]}
    assert edges["gen-client"]["dropout_rate"] == 0.01          # NetworkParameters.DROPOUT_RATE
    assert edges["lb-srv2"]["dropout_rate"] == 0.0
    assert edges["client-lb"]["latency"]["variance"] == 0.002   # variance defaults to mean (log_normal)
    assert edges["gen-client"]["latency"]["variance"] is None
    assert p["sim_settings"]["sample_period_s"] == 0.01
    assert len(p["sim_settings"]["enabled_sample_metrics"]) == 4
    assert p["topology_graph"]["nodes"]["load_balancer"]["algorithms"] == "round_robin"
    assert p["topology_graph"]["nodes"]["servers"][0]["endpoints"][0]["endpoint_name"] == "/api"
    assert normalize_payload(p) == p  # idempotent


def test_lowering_wires_like_the_runner():
    plan = lower(lb_two_servers())
    assert plan.edge_ids == ["gen-client", "client-lb", "lb-srv1", "lb-srv2", "srv1-client", "srv2-client"]
    assert plan.gen_out_edge == 0 and plan.client_out_edge == 1
    assert list(plan.lb_edges) == [2, 3] and list(plan.srv_out_edge) == [4, 5]
    assert list(plan.edge_target_kind) == [0, 1, 2, 2, 0, 0] and list(plan.edge_target_idx[2:4]) == [0, 1]
    # RAM steps folded into ep_ram, CPU/IO steps kept in order
    assert list(plan.ep_ram) == [128.0, 128.0]
    assert list(plan.step_kind) == [0, 1, 0, 1] and list(plan.ep_step_begin) == [0, 2, 4]
    assert plan.metrics_mask == 15 and plan.n_series == 12 and plan.tick_count == 11999
    assert plan.step_index[(0, 0, 1)] == -1 and plan.step_index[(1, 0, 2)] == 3


def test_event_timelines_sorted_end_before_start():
    p = lb_with_events(horizon=600)
    p["events"].append({"event_id": "a-back-to-back", "target_id": "client-lb",
                        "start": {"kind": "network_spike_start", "t_start": 160.0, "spike_s": 0.5},
                        "end": {"kind": "network_spike_end", "t_end": 170.0}})
    plan = lower(p)
    t, d = list(plan.emark_time), list(plan.emark_delta)
    assert t == sorted(t)
    i = t.index(160.0)
    assert d[i] == -0.015 and d[i + 1] == 0.5       # END before START at equal time (injection.py:142-151)
    assert list(plan.smark_down) == [1, 0, 1, 0] and list(plan.smark_lb_edge) == [2, 2, 3, 3]


def test_mark_times_follow_relative_wait_accumulation():
    p = single_server(horizon=10)
    p["events"] = [
        {"event_id": "e1", "target_id": "gen-to-client", "start": {"kind": "network_spike_start", "t_start": 0.1, "spike_s": 0.01},
         "end": {"kind": "network_spike_end", "t_end": 0.3}},
        {"event_id": "e2", "target_id": "gen-to-client", "start": {"kind": "network_spike_start", "t_start": 0.7, "spike_s": 0.01},
         "end": {"kind": "network_spike_end", "t_end": 1.1}},
    ]
    plan = lower(p)
    now, last, want = 0.0, 0.0, []
    for t in (0.1, 0.3, 0.7, 1.1):
        now, last = now + (t - last), t
        want.append(now)
    assert list(plan.emark_time) == want


@pytest.mark.parametrize("mutate,match", [
    (lambda p: p["topology_graph"]["edges"][0]["latency"].update(mean=0), "must be positive"),
    (lambda p: p["topology_graph"]["edges"][0].update(target="nope"), "unknown target"),
    (lambda p: p["topology_graph"]["edges"].append(dict(p["topology_graph"]["edges"][0])), "multiple edges"),
    (lambda p: p["rqs_input"]["avg_active_users"].update(distribution="uniform"), "Poisson or Gaussian"),
    (lambda p: p["rqs_input"]["avg_request_per_minute_per_user"].update(distribution="normal"), "must be Poisson"),
    (lambda p: p["sim_settings"].update(total_simulation_time=3), ">= 5"),
    (lambda p: p["sim_settings"].update(sample_period_s=0.5), "sample_period_s"),
    (lambda p: p["topology_graph"]["nodes"]["servers"][0]["endpoints"][0]["steps"][0].update(step_operation={"io_waiting_time": 1}), "must use cpu_time"),
    (lambda p: p["topology_graph"]["nodes"]["load_balancer"].update(server_covered=["srv-9"]), "unknown servers"),
    (lambda p: p.update(events=[{"event_id": "x", "target_id": "srv-1", "start": {"kind": "server_down", "t_start": 1.0},
                                 "end": {"kind": "server_up", "t_end": 700.0}}]), "horizon"),
    (lambda p: p.update(events=[
        {"event_id": "x", "target_id": "srv-1", "start": {"kind": "server_down", "t_start": 1.0}, "end": {"kind": "server_up", "t_end": 9.0}},
        {"event_id": "y", "target_id": "srv-2", "start": {"kind": "server_down", "t_start": 2.0}, "end": {"kind": "server_up", "t_end": 5.0}}]),
     "all servers are down"),
    (lambda p: p.update(events=[{"event_id": "x", "target_id": "client-lb", "start": {"kind": "network_spike_start", "t_start": 1.0},
                                 "end": {"kind": "network_spike_end", "t_end": 2.0}}]), "spike_s"),
])
def test_invalid_payloads_raise_value_error(mutate, match):
    p = copy.deepcopy(lb_two_servers(horizon=600))
    mutate(p)
    with pytest.raises(ValueError, match=match):
        lower(p)


def test_servers_without_endpoints_are_rejected():
    p = lb_two_servers()
    p["topology_graph"]["nodes"]["servers"][0]["endpoints"] = []
    with pytest.raises(ValueError, match="no endpoints"):
        lower(p)


def test_sweep_paths_resolve_to_override_columns():
    plan = lower(lb_two_servers())
    n = 4
    ov = resolve_sweep(plan, {
        "rqs_input.avg_active_users.mean": [10, 20, 30, 40],
        "topology_graph.edges[*].latency.mean": 0.004,
        "topology_graph.edges[lb-srv2].dropout_rate": [0, 0, 0.5, 0.5],
        "topology_graph.nodes.servers[srv-2].endpoints[0].steps[2].io_waiting_time": 0.05,
    }, n)
    codes = [(c, i) for c, i, _, _ in ov]
    assert (_abi.PARAM_CODES["gen_users_mean"], 0) in codes
    assert sum(1 for c, _ in codes if c == _abi.PARAM_CODES["edge_mean"]) == 6
    assert (_abi.PARAM_CODES["edge_dropout"], 3) in codes and (_abi.PARAM_CODES["step_time"], 3) in codes
    assert all(v.shape == (n,) for _, _, v, _ in ov)
    with pytest.raises(ValueError):
        resolve_sweep(plan, {"topology_graph.edges[zzz].latency.mean": 1.0}, n)
    with pytest.raises(ValueError):
        resolve_sweep(plan, {"topology_graph.nodes.servers[srv-1].endpoints[0].steps[1].cpu_time": 1.0}, n)  # a RAM step


def test_capacity_estimates_cover_observed_live_requests():
    from oracle import oracle_lib as ol

    for payload, seed in ((lb_two_servers(horizon=60), 1), (stress_mixed(30), 3)):
        plan = lower(payload)
        cap, fifo = estimate_capacities(plan)
        live = int(ol.simulate(plan, seed).counts[_abi.CNT_MAX_LIVE])
        assert cap >= live and fifo >= 8 and (fifo & (fifo - 1)) == 0
    plan = lower(lb_two_servers(horizon=600))
    mean, std = plan.expected_arrivals()
    assert abs(mean - 80000) < 1 and 1000 < std < 2000
    assert plan.clock_capacity() > mean + 6 * std


def test_an_outage_that_empties_the_load_balancer_is_rejected_at_lowering():
    """The LB covers srv-1 only (srv-2 is wired to the client but receives nothing): taking srv-1 down
    passes the reference's "not all servers down" validator but leaves the LB with nothing to route
    to -- the reference fails inside the run (StopIteration, lb_algorithms.py:33); lowering refuses."""
    import copy

    import pytest

    from asyncflow_amd.plan import lower
    from asyncflow_amd.workloads import lb_two_servers

    p = copy.deepcopy(lb_two_servers(horizon=30))
    p["topology_graph"]["nodes"]["load_balancer"]["server_covered"] = ["srv-1"]
    p["topology_graph"]["edges"] = [e for e in p["topology_graph"]["edges"] if e["id"] != "lb-srv2"]
    lower(p)                                                       # fine without the outage
    p["events"] = [{"event_id": "down", "target_id": "srv-1", "start": {"kind": "server_down", "t_start": 5.0},
                    "end": {"kind": "server_up", "t_end": 9.0}}]
    with pytest.raises(ValueError, match="no out-edge to route to"):
        lower(p)
