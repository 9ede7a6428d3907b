"""BASELINE.json workloads as payload dicts (the values of the reference's example YAMLs).

``single_server``  examples/yaml_input/data/single_server.yml            (config 1)
``lb_two_servers`` examples/yaml_input/data/two_servers_lb.yml:14-71      (config 2, "LB-2")
``lb_with_events`` LB-2 + the events of examples/yaml_input/data/event_inj_lb.yml:73-102 (config 4)
``single_server_with_spike`` examples/yaml_input/data/event_inj_single_server.yml / heavy_inj_single_server.yml
                   (a 2 s / 3 s network spike on the client -> server edge: rate x spike messages pile up when it ends)
``fanout8``        8-server fan-out with log-normal edges     (config 5)
``grid_users_rtt`` the users x RTT grid of configs 3 / 4 as sweep columns
``lb_two_servers_two_endpoints`` LB-2 with two endpoints per server and core re-entry (general servers; bench --config 6)

Pure data: no engine and no oracle code.  bench.py, the tests and the oracle's scenario
library all take the BASELINE workloads from here
(tests/test_reference_live.py::test_baseline_workloads_equal_the_reference_yaml pins them on
the reference's YAML files).
"""

from __future__ import annotations

from typing import Any

import numpy as np


def _server(sid: str, cores: int = 1, ram: int = 2048, endpoints: list | None = None) -> dict:
    if endpoints is None:
        endpoints = [_endpoint("/api", [("initial_parsing", 0.002), ("ram", 128), ("io_wait", 0.012)])]
    return {"id": sid, "server_resources": {"cpu_cores": cores, "ram_mb": ram}, "endpoints": endpoints}


def _endpoint(name: str, steps: list[tuple[str, float]]) -> dict:
    out = []
    for kind, val in steps:
        key = "necessary_ram" if kind == "ram" else ("cpu_time" if kind in ("initial_parsing", "cpu_bound_operation") else "io_waiting_time")
        out.append({"kind": kind, "step_operation": {key: val}})
    return {"endpoint_name": name, "steps": out}


def _edge(eid: str, src: str, tgt: str, mean: float, dist: str = "exponential", variance: float | None = None, dropout: float | None = None) -> dict:
    lat: dict[str, Any] = {"mean": mean, "distribution": dist}
    if variance is not None:
        lat["variance"] = variance
    e: dict[str, Any] = {"id": eid, "source": src, "target": tgt, "latency": lat}
    if dropout is not None:
        e["dropout_rate"] = dropout
    return e


def single_server(users: float = 100, rpm: float = 20, horizon: int = 300, period: float = 0.05) -> dict:
    """examples/yaml_input/data/single_server.yml (BASELINE config 1; the file says T=500)."""
    return {
        "rqs_input": {
            "id": "rqs-1",
            "avg_active_users": {"mean": users},
            "avg_request_per_minute_per_user": {"mean": rpm},
            "user_sampling_window": 60,
        },
        "topology_graph": {
            "nodes": {
                "client": {"id": "client-1"},
                "servers": [_server("srv-1", 1, 2048, [_endpoint("ep-1", [("initial_parsing", 0.001), ("ram", 100), ("io_wait", 0.1)])])],
            },
            "edges": [
                _edge("gen-to-client", "rqs-1", "client-1", 0.003),
                _edge("client-to-server", "client-1", "srv-1", 0.003),
                _edge("server-to-client", "srv-1", "client-1", 0.003),
            ],
        },
        "sim_settings": {"total_simulation_time": horizon, "sample_period_s": period},
    }


def lb_two_servers(users: float = 400, rpm: float = 20, horizon: int = 600, period: float = 0.05, algo: str = "round_robin") -> dict:
    """examples/yaml_input/data/two_servers_lb.yml:14-71 (BASELINE config 2, "LB-2")."""
    return {
        "rqs_input": {
            "id": "rqs-1",
            "avg_active_users": {"mean": users},
            "avg_request_per_minute_per_user": {"mean": rpm},
            "user_sampling_window": 60,
        },
        "topology_graph": {
            "nodes": {
                "client": {"id": "client-1"},
                "load_balancer": {"id": "lb-1", "algorithms": algo, "server_covered": ["srv-1", "srv-2"]},
                "servers": [_server("srv-1"), _server("srv-2")],
            },
            "edges": [
                _edge("gen-client", "rqs-1", "client-1", 0.003),
                _edge("client-lb", "client-1", "lb-1", 0.002),
                _edge("lb-srv1", "lb-1", "srv-1", 0.002),
                _edge("lb-srv2", "lb-1", "srv-2", 0.002),
                _edge("srv1-client", "srv-1", "client-1", 0.003),
                _edge("srv2-client", "srv-2", "client-1", 0.003),
            ],
        },
        "sim_settings": {
            "total_simulation_time": horizon,
            "sample_period_s": period,
            "enabled_sample_metrics": ["ready_queue_len", "event_loop_io_sleep", "ram_in_use", "edge_concurrent_connection"],
            "enabled_event_metrics": ["rqs_clock"],
        },
    }


def lb_two_servers_two_endpoints(users: float = 400, horizon: int = 600) -> dict:
    """LB-2 with a second endpoint on both servers whose step program comes BACK to the core queue after an I/O step
    (io_db 4 ms, 64 MB, cpu 1.5 ms, io_wait 6 ms, cpu 0.5 ms): server.py:79-313 without the tandem restriction -- one
    uniform endpoint draw per arrival (server.py:101), RAM needs that differ per request.  Not a BASELINE config: the
    workload the general server station of the stage-parallel kernel is measured on (`bench.py --config 6`)."""
    p = lb_two_servers(users=users, horizon=horizon)
    for s in p["topology_graph"]["nodes"]["servers"]:
        s["endpoints"].append(_endpoint("/report", [("io_db", 0.004), ("ram", 64), ("cpu_bound_operation", 0.0015),
                                                    ("io_wait", 0.006), ("cpu_bound_operation", 0.0005)]))
    return p


def lb_with_events(users: float = 120, horizon: int = 600, scale: float = 1.0) -> dict:
    """examples/yaml_input/data/event_inj_lb.yml:73-102 (BASELINE config 4's events).

    ``scale`` compresses the event times so short-horizon fixtures still see them.
    """
    p = lb_two_servers(users=users, horizon=horizon)
    s = scale
    p["events"] = [
        {"event_id": "ev-spike-1", "target_id": "client-lb",
         "start": {"kind": "network_spike_start", "t_start": 100.0 * s, "spike_s": 0.015}, "end": {"kind": "network_spike_end", "t_end": 160.0 * s}},
        {"event_id": "ev-srv1-down", "target_id": "srv-1",
         "start": {"kind": "server_down", "t_start": 180.0 * s}, "end": {"kind": "server_up", "t_end": 240.0 * s}},
        {"event_id": "ev-spike-2", "target_id": "lb-srv2",
         "start": {"kind": "network_spike_start", "t_start": 300.0 * s, "spike_s": 0.020}, "end": {"kind": "network_spike_end", "t_end": 360.0 * s}},
        {"event_id": "ev-srv2-down", "target_id": "srv-2",
         "start": {"kind": "server_down", "t_start": 360.0 * s}, "end": {"kind": "server_up", "t_end": 420.0 * s}},
        {"event_id": "ev-spike-3", "target_id": "gen-client",
         "start": {"kind": "network_spike_start", "t_start": 480.0 * s, "spike_s": 0.010}, "end": {"kind": "network_spike_end", "t_end": 540.0 * s}},
    ]
    return p


def single_server_with_spike(heavy: bool = False, horizon: int | None = None, scale: float = 1.0) -> dict:
    """examples/yaml_input/data/event_inj_single_server.yml, or heavy_inj_single_server.yml with ``heavy``.

    ``scale`` compresses the event times (and ``horizon`` the run) so short fixtures still see the spike end.
    """
    if heavy:
        p = single_server(users=300, rpm=30, horizon=600 if horizon is None else horizon)
        srv = p["topology_graph"]["nodes"]["servers"][0]
        srv["server_resources"]["ram_mb"] = 8000
        srv["endpoints"] = [_endpoint("ep-1", [("initial_parsing", 0.005), ("ram", 200), ("io_wait", 0.2)])]
        ev = {"event_id": "ev-spike-heavy", "target_id": "client-to-server",
              "start": {"kind": "network_spike_start", "t_start": 180.0 * scale, "spike_s": 3.0}, "end": {"kind": "network_spike_end", "t_end": 300.0 * scale}}
    else:
        p = single_server(users=100, rpm=20, horizon=500 if horizon is None else horizon)
        ev = {"event_id": "ev-spike-1", "target_id": "client-to-server",
              "start": {"kind": "network_spike_start", "t_start": 120.0 * scale, "spike_s": 2.0}, "end": {"kind": "network_spike_end", "t_end": 240.0 * scale}}
    p["sim_settings"]["enabled_sample_metrics"] = ["ready_queue_len", "event_loop_io_sleep", "ram_in_use", "edge_concurrent_connection"]
    p["sim_settings"]["enabled_event_metrics"] = ["rqs_clock"]
    p["events"] = [ev]
    return p


def fanout8(users: float = 120, horizon: int = 600, period: float = 0.05) -> dict:
    """BASELINE config 5: LB -> 8 identical servers, log-normal edges."""
    servers = [_server(f"srv-{i}") for i in range(1, 9)]
    edges = [
        _edge("gen-client", "rqs-1", "client-1", 0.001, "log_normal", 0.25),
        _edge("client-lb", "client-1", "lb-1", 0.001, "log_normal", 0.25),
    ]
    for i in range(1, 9):
        edges.append(_edge(f"lb-srv{i}", "lb-1", f"srv-{i}", 0.001, "log_normal", 0.25))
        edges.append(_edge(f"srv{i}-client", f"srv-{i}", "client-1", 0.001, "log_normal", 0.25))
    return {
        "rqs_input": {"id": "rqs-1", "avg_active_users": {"mean": users}, "avg_request_per_minute_per_user": {"mean": 20}, "user_sampling_window": 60},
        "topology_graph": {
            "nodes": {
                "client": {"id": "client-1"},
                "load_balancer": {"id": "lb-1", "algorithms": "round_robin", "server_covered": [f"srv-{i}" for i in range(1, 9)]},
                "servers": servers,
            },
            "edges": edges,
        },
        "sim_settings": {"total_simulation_time": horizon, "sample_period_s": period},
    }


def grid_users_rtt(side: int = 100, users_max: float = 1000.0, hop_mean_max: float = 0.05) -> tuple[np.ndarray, np.ndarray]:
    """``avg_active_users.mean = users_max/side * a`` (a = 1..side) x per-hop edge
    latency mean ``= hop_mean_max/side * b`` (b = 1..side, applied to every edge); returns the two
    columns of the side*side scenarios in row-major (a, b) order."""
    a = np.repeat(np.arange(1, side + 1), side) * (users_max / side)
    b = np.tile(np.arange(1, side + 1), side) * (hop_mean_max / side)
    return a.astype(np.float64), b.astype(np.float64)


BASELINE_SEED_BASE = {1: 0, 2: 0x5EED0000, 3: 0xC0F30000, 4: 0xC0F40000, 5: 0xFA085000, 6: 0x5EED0000}


def reference_examples() -> dict[str, dict]:
    """The reference's five example inputs (examples/yaml_input/data/*.yml), value for value (pinned on the YAML files by
    tests/test_reference_live.py::test_baseline_workloads_equal_the_reference_yaml) -- plus the README quickstart
    (README.md:109-155: 100 users x 100 rpm, 300 s).  `SimulationRunner.prebuild_reference_examples()` builds their
    plan-specialised kernels where a compiler is, so that a user who runs the reference's own examples on a box without one
    gets the specialised kernel on the first sweep."""
    return {
        "single_server.yml": single_server(horizon=500),
        "two_servers_lb.yml": lb_two_servers(),
        "event_inj_lb.yml": lb_with_events(users=120, horizon=600),
        "event_inj_single_server.yml": single_server_with_spike(),
        "heavy_inj_single_server.yml": single_server_with_spike(heavy=True),
        "README quickstart": single_server(users=100, rpm=100, horizon=300),
    }
