"""Input normalisation for the batched engine: the drop-in surface.

The engine keeps AsyncFlow's YAML / Pydantic input schema unchanged
(/root/reference/src/asyncflow/schemas/payload.py:12-18).  Two entry forms:

* a validated reference ``SimulationPayload`` (any pydantic model exposing
  ``model_dump``) -- used as is, the reference's own validators already ran;
* a plain ``dict`` (``yaml.safe_load`` output).  If the reference package
  ``asyncflow`` is importable it is validated by the reference's own models
  (identical error behaviour); otherwise by the structural checks below, which
  mirror the defaults and the validators that guard the *hot path* (citations
  inline).  They raise ``ValueError`` like pydantic's ``ValidationError`` does
  (``ValidationError`` is a ``ValueError`` subclass).

The result is a plain-JSON ``dict`` with every default filled in; lowering to
the device plan happens in :mod:`asyncflow_amd.plan`.
"""

from __future__ import annotations

import copy
from collections import Counter
from pathlib import Path
from typing import Any, Mapping

DISTRIBUTIONS = ("poisson", "normal", "log_normal", "exponential", "uniform")
CPU_STEP_KINDS = ("initial_parsing", "cpu_bound_operation")
IO_STEP_KINDS = ("io_task_spawn", "io_llm", "io_wait", "io_db", "io_cache")
RAM_STEP_KINDS = ("ram",)
SAMPLED_METRICS = (
    "ready_queue_len",
    "event_loop_io_sleep",
    "ram_in_use",
    "edge_concurrent_connection",
)
LB_ALGORITHMS = ("round_robin", "least_connection")


def _fail(msg: str) -> None:
    raise ValueError(msg)


def _enum(value: Any, allowed: tuple[str, ...], what: str) -> str:
    v = getattr(value, "value", value)
    if not isinstance(v, str) or v not in allowed:
        _fail(f"{what}: {value!r} is not one of {list(allowed)}")
    return v


def _num(value: Any, what: str) -> float:
    if isinstance(value, bool) or not isinstance(value, (int, float)):
        _fail(f"{what} must be a number (int or float)")
    return float(value)


def _rv(d: Any, what: str) -> dict:
    """RVConfig (schemas/common/random_variables.py:8-37)."""
    if not isinstance(d, Mapping):
        _fail(f"{what} must be a mapping with at least 'mean'")
    if "mean" not in d:
        _fail(f"{what}: field 'mean' is required")
    mean = _num(d["mean"], f"{what}.mean")
    dist = _enum(d.get("distribution", "poisson"), DISTRIBUTIONS, f"{what}.distribution")
    var = d.get("variance")
    if var is not None:
        var = _num(var, f"{what}.variance")
    elif dist in ("normal", "log_normal"):
        var = mean  # random_variables.py:27-37
    return {"mean": mean, "distribution": dist, "variance": var}


def _normalize_step(d: Any, where: str) -> dict:
    """Step (schemas/topology/endpoint.py:19-88)."""
    kind = _enum(d.get("kind"), CPU_STEP_KINDS + IO_STEP_KINDS + RAM_STEP_KINDS, f"{where}.kind")
    op = d.get("step_operation")
    if not op:
        _fail(f"{where}: step_operation cannot be empty")
    op = {getattr(k, "value", k): v for k, v in dict(op).items()}
    if len(op) != 1:
        _fail(f"{where}: step_operation must contain exactly one entry")
    want = "cpu_time" if kind in CPU_STEP_KINDS else ("necessary_ram" if kind in RAM_STEP_KINDS else "io_waiting_time")
    if set(op) != {want}:
        _fail(f"{where}: a {kind} step must use {want}")
    val = op[want]
    if isinstance(val, bool) or not isinstance(val, (int, float)) or not val > 0:
        _fail(f"{where}: {want} must be a positive number")
    return {"kind": kind, "step_operation": {want: val}}


def _normalize(data: Mapping[str, Any]) -> dict:  # noqa: C901, PLR0912, PLR0915
    for key in ("rqs_input", "topology_graph", "sim_settings"):
        if key not in data:
            _fail(f"payload: field '{key}' is required")

    # -- RqsGenerator (schemas/workload/rqs_generator.py:10-59)
    g = data["rqs_input"]
    users = _rv(g.get("avg_active_users"), "rqs_input.avg_active_users")
    rpm = _rv(g.get("avg_request_per_minute_per_user"), "rqs_input.avg_request_per_minute_per_user")
    if rpm["distribution"] != "poisson":
        _fail("At the moment the variable avg request must be Poisson")
    if users["distribution"] not in ("poisson", "normal"):
        _fail("At the moment the variable active user must be Poisson or Gaussian")
    window = g.get("user_sampling_window", 60)
    if isinstance(window, bool) or not isinstance(window, int) or not 1 <= window <= 120:
        _fail("rqs_input.user_sampling_window must be an integer in [1, 120]")
    rqs = {
        "id": str(g["id"]),
        "type": "generator",
        "avg_active_users": users,
        "avg_request_per_minute_per_user": rpm,
        "user_sampling_window": window,
    }

    # -- SimulationSettings (schemas/settings/simulation.py:13-44)
    s = data["sim_settings"] or {}
    T = s.get("total_simulation_time", 3600)
    if isinstance(T, float) and T.is_integer():
        T = int(T)
    if isinstance(T, bool) or not isinstance(T, int) or T < 5:
        _fail("sim_settings.total_simulation_time must be an integer >= 5")
    period = _num(s.get("sample_period_s", 0.01), "sim_settings.sample_period_s")
    if not 0.001 <= period <= 0.1:
        _fail("sim_settings.sample_period_s must be within [0.001, 0.1]")
    metrics = s.get("enabled_sample_metrics")
    metrics = list(SAMPLED_METRICS) if metrics is None else [
        _enum(m, SAMPLED_METRICS, "sim_settings.enabled_sample_metrics") for m in metrics
    ]
    ev_metrics = s.get("enabled_event_metrics")
    ev_metrics = ["rqs_clock"] if ev_metrics is None else [
        _enum(m, ("rqs_clock", "llm_cost"), "sim_settings.enabled_event_metrics") for m in ev_metrics
    ]
    settings = {
        "total_simulation_time": T,
        "enabled_sample_metrics": sorted(set(metrics)),
        "enabled_event_metrics": sorted(set(ev_metrics)),
        "sample_period_s": period,
    }

    # -- TopologyNodes (schemas/topology/nodes.py:34-166)
    tg = data["topology_graph"]
    nodes = tg.get("nodes") or {}
    extra = set(nodes) - {"servers", "client", "load_balancer"}
    if extra:
        _fail(f"topology_graph.nodes: unknown fields {sorted(extra)}")  # extra="forbid"
    if "client" not in nodes or "servers" not in nodes:
        _fail("topology_graph.nodes needs 'client' and 'servers'")
    client = {"id": str(nodes["client"]["id"]), "type": "client"}
    servers = []
    for i, sv in enumerate(nodes["servers"]):
        res = dict(sv.get("server_resources") or {})
        cores = res.get("cpu_cores", 1)
        ram = res.get("ram_mb", 1024)
        if isinstance(cores, bool) or not isinstance(cores, int) or cores < 1:
            _fail(f"servers[{i}].server_resources.cpu_cores must be an integer >= 1")
        if isinstance(ram, bool) or not isinstance(ram, int) or ram < 256:
            _fail(f"servers[{i}].server_resources.ram_mb must be an integer >= 256")
        eps = []
        for j, ep in enumerate(sv.get("endpoints") or []):
            steps = [
                _normalize_step(st, f"servers[{i}].endpoints[{j}].steps[{k}]")
                for k, st in enumerate(ep.get("steps") or [])
            ]
            eps.append({"endpoint_name": str(ep["endpoint_name"]).lower(), "steps": steps})
        servers.append({
            "id": str(sv["id"]),
            "type": "server",
            "server_resources": {
                "cpu_cores": cores,
                "db_connection_pool": res.get("db_connection_pool"),
                "ram_mb": ram,
            },
            "endpoints": eps,
        })
    lb = None
    if nodes.get("load_balancer") is not None:
        lbd = nodes["load_balancer"]
        lb = {
            "id": str(lbd["id"]),
            "type": "load_balancer",
            "algorithms": _enum(lbd.get("algorithms", "round_robin"), LB_ALGORITHMS, "load_balancer.algorithms"),
            "server_covered": sorted({str(x) for x in (lbd.get("server_covered") or [])}),
        }
    ids = [sv["id"] for sv in servers] + [client["id"]] + ([lb["id"]] if lb else [])
    dup = [k for k, v in Counter(ids).items() if v > 1]
    if dup:
        _fail(f"The following node ids are duplicate {dup}")  # nodes.py:147-164

    # -- Edges (schemas/topology/edges.py:25-97) + graph rules (graph.py:24-159)
    edges = []
    for i, e in enumerate(tg.get("edges") or []):
        lat = _rv(e.get("latency"), f"edges[{i}].latency")
        if lat["mean"] <= 0:
            _fail(f"The mean latency of the edge '{e.get('id', 'unknown')}' must be positive")
        if lat["variance"] is not None and lat["variance"] < 0:
            _fail(f"The variance of the latency of the edge {e.get('id', 'unknown')} must be non negative")
        drop = _num(e.get("dropout_rate", 0.01), f"edges[{i}].dropout_rate")
        if not 0.0 <= drop <= 1.0:
            _fail(f"edges[{i}].dropout_rate must be within [0, 1]")
        if str(e["source"]) == str(e["target"]):
            _fail("source and target must be different nodes")
        edges.append({
            "id": str(e["id"]),
            "source": str(e["source"]),
            "target": str(e["target"]),
            "latency": lat,
            "edge_type": "network_connection",
            "dropout_rate": drop,
        })
    dup = [k for k, v in Counter(e["id"] for e in edges).items() if v > 1]
    if dup:
        _fail(f"There are multiple edges with the following ids {dup}")
    node_ids = set(ids)
    external = set()
    for e in edges:
        if e["target"] not in node_ids:
            _fail(f"Edge {e['source']}->{e['target']} references unknown target node '{e['target']}'.")
        if e["source"] not in node_ids:
            external.add(e["source"])
    bad = external & {e["target"] for e in edges}
    if bad:
        _fail(f"External IDs cannot be used as targets as well:{sorted(bad)}")
    if lb is not None:
        server_ids = {sv["id"] for sv in servers}
        missing = set(lb["server_covered"]) - server_ids
        if missing:
            _fail(f"Load balancer '{lb['id']}'references unknown servers: {sorted(missing)}")
        linked = {e["target"] for e in edges if e["source"] == lb["id"]}
        not_linked = set(lb["server_covered"]) - linked
        if not_linked:
            _fail(f"Servers {sorted(not_linked)} are covered by LB '{lb['id']}' but have no outgoing edge from it.")
    fan: dict[str, int] = {}
    for e in edges:
        if e["source"] in node_ids:
            fan[e["source"]] = fan.get(e["source"], 0) + 1
    offenders = [k for k, c in fan.items() if c > 1 and (lb is None or k != lb["id"])]
    if offenders:
        _fail(f"Only the load balancer can have multiple outgoing edges. Offending sources: {offenders}")

    # -- EventInjection (schemas/events/injection.py:25-119, payload.py:20-252)
    events = None
    if data.get("events"):
        events = []
        T_f = float(T)
        server_ids = {sv["id"] for sv in servers}
        edge_ids = {e["id"] for e in edges}
        for ev in data["events"]:
            st, en = dict(ev["start"]), dict(ev["end"])
            if set(st) - {"kind", "t_start", "spike_s"} or set(en) - {"kind", "t_end"}:
                _fail(f"Event {ev.get('event_id')}: unknown fields in start/end")
            sk = _enum(st.get("kind"), ("server_down", "network_spike_start"), "event.start.kind")
            ek = _enum(en.get("kind"), ("server_up", "network_spike_end"), "event.end.kind")
            t0 = _num(st.get("t_start"), "event.start.t_start")
            t1 = _num(en.get("t_end"), "event.end.t_end")
            spike = st.get("spike_s")
            eid, tgt = str(ev["event_id"]), str(ev["target_id"])
            if t0 < 0 or not t1 > 0:
                _fail(f"Event '{eid}': t_start must be >= 0 and t_end > 0")
            want = {"server_down": "server_up", "network_spike_start": "network_spike_end"}[sk]
            if ek != want:
                _fail(f"The event {eid} must have as value of kind in end {want}")
            if t0 >= t1:
                _fail(f"The starting time for the event {eid} must be smaller than the ending time")
            if sk == "network_spike_start":
                if spike is None or not _num(spike, "spike_s") > 0:
                    _fail(f"The field spike_s for the event {eid} must be defined as a positive float")
                spike = float(spike)
            elif spike is not None:
                _fail(f"Event {eid}: spike_s must be omitted")
            if tgt not in server_ids | edge_ids:
                _fail(f"The target id {tgt} related to the event {eid} does not exist")
            if t0 > T_f or t1 > T_f:
                _fail(f"Event '{eid}': window [{t0:.6f}, {t1:.6f}] exceeds simulation horizon T={T_f:.6f}")
            if sk == "server_down" and tgt not in server_ids:
                _fail(f"The event {eid} regarding a server does not have a compatible target id")
            if sk == "network_spike_start" and tgt not in edge_ids:
                _fail(f"The event {eid} regarding an edge does not have a compatible target id")
            events.append({
                "event_id": eid,
                "target_id": tgt,
                "start": {"kind": sk, "t_start": t0, "spike_s": spike},
                "end": {"kind": ek, "t_end": t1},
            })
        if len({e["event_id"] for e in events}) != len(events):
            _fail("The id's representing different events must be unique")
        # payload.py:146-202 / 204-252: never all servers down, no overlapping outages
        timeline = []
        for e in events:
            if e["target_id"] in server_ids:
                timeline.append((e["start"]["t_start"], "start", e["target_id"]))
                timeline.append((e["end"]["t_end"], "end", e["target_id"]))
        timeline.sort(key=lambda x: (x[0], x[1] == "start"))
        down: set[str] = set()
        for t, kind, sid in timeline:
            if kind == "end":
                down.discard(sid)
            else:
                if sid in down:
                    _fail(f"Overlapping events for server '{sid}' at t={t:.6f}; server outage windows must not overlap.")
                down.add(sid)
                if len(down) == len(server_ids):
                    _fail(f"At time {t:.6f} all servers are down; keep at least one up")

    return {
        "rqs_input": rqs,
        "topology_graph": {
            "nodes": {"servers": servers, "client": client, "load_balancer": lb},
            "edges": edges,
        },
        "sim_settings": settings,
        "events": events,
    }


def _reference_model():
    try:
        from asyncflow.schemas.payload import SimulationPayload  # type: ignore[import-not-found]
    except Exception:  # noqa: BLE001 - reference package absent (e.g. on the GPU box)
        return None
    return SimulationPayload


def normalize_payload(payload: Any) -> dict:
    """Return the payload as a plain dict with all defaults applied.

    Raises ``ValueError`` (pydantic's ``ValidationError`` is one) on invalid input.
    """
    if hasattr(payload, "model_dump"):
        return _normalize(payload.model_dump(mode="json"))
    if not isinstance(payload, Mapping):
        msg = f"unsupported payload type {type(payload)!r}"
        raise TypeError(msg)
    model = _reference_model()
    if model is not None:
        return _normalize(model.model_validate(copy.deepcopy(dict(payload))).model_dump(mode="json"))
    return _normalize(payload)


def load_yaml(yaml_path: str | Path) -> dict:
    """``SimulationRunner.from_yaml`` front half (simulation_runner.py:396-398)."""
    import yaml

    return normalize_payload(yaml.safe_load(Path(yaml_path).read_text()))
