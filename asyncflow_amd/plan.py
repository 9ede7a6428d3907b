"""Lower a normalised payload to the flat device plan (``af_plan_t``).

This is the batched analogue of the reference's per-run build/wire phase
(/root/reference/src/asyncflow/runtime/simulation_runner.py:127-294): it is done
ONCE per sweep, the plan is shared by every scenario, and per-scenario
parameters travel as override columns (``af_sweep_t``).
"""

from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass, field
from typing import Any

import numpy as np

from . import _abi
from .payload import CPU_STEP_KINDS, IO_STEP_KINDS, RAM_STEP_KINDS, normalize_payload


def _ptr(arr: np.ndarray, ctype: Any) -> Any:
    return arr.ctypes.data_as(C.POINTER(ctype))


@dataclass
class DevicePlan:
    """Flat SoA description of one topology + workload + event timelines."""

    payload: dict
    total_time: float
    sample_period: float
    metrics_mask: int
    gen_users_dist: int
    gen_users_mean: float
    gen_users_sigma: float
    gen_rpm_mean: float
    gen_window_s: float
    gen_out_edge: int
    client_out_edge: int
    has_lb: bool
    lb_algo: int
    lb_edges: np.ndarray
    edge_target_kind: np.ndarray
    edge_target_idx: np.ndarray
    edge_dist: np.ndarray
    edge_mean: np.ndarray
    edge_sigma: np.ndarray
    edge_dropout: np.ndarray
    srv_cores: np.ndarray
    srv_ram_mb: np.ndarray
    srv_out_edge: np.ndarray
    srv_ep_begin: np.ndarray
    ep_step_begin: np.ndarray
    ep_ram: np.ndarray
    step_kind: np.ndarray
    step_time: np.ndarray
    emark_time: np.ndarray
    emark_edge: np.ndarray
    emark_delta: np.ndarray
    smark_time: np.ndarray
    smark_lb_edge: np.ndarray
    smark_down: np.ndarray
    edge_ids: list[str] = field(default_factory=list)
    server_ids: list[str] = field(default_factory=list)
    #: payload step (server, endpoint, step) -> flat step index (RAM steps: -1)
    step_index: dict[tuple[int, int, int], int] = field(default_factory=dict)

    @property
    def n_edges(self) -> int:
        return len(self.edge_ids)

    @property
    def n_servers(self) -> int:
        return len(self.server_ids)

    @property
    def n_series(self) -> int:
        return self.n_edges + 3 * self.n_servers

    @property
    def series_pitch(self) -> int:
        """Row length of the device sample array: ``n_series`` rounded up to a multiple of 4."""
        return (self.n_series + 3) & ~3

    @property
    def tick_count(self) -> int:
        """Ticks ``env.run(until=T)`` takes (metrics/collector.py:50-53).

        The collector's clock is ``0 + p + p + ...`` in f64 and the stop event
        is URGENT at T, so the count follows repeated addition, not ``T/p``.
        """
        n, t = 0, 0.0 + self.sample_period
        while t < self.total_time:
            n += 1
            t = t + self.sample_period
        return n

    def as_ctypes(self) -> _abi.AfPlan:
        p = _abi.AfPlan()
        p.abi_version = _abi.AF_ABI_VERSION
        p.struct_size = C.sizeof(_abi.AfPlan)
        p.total_time = self.total_time
        p.sample_period = self.sample_period
        p.metrics_mask = self.metrics_mask
        p.gen_users_dist = self.gen_users_dist
        p.gen_users_mean = self.gen_users_mean
        p.gen_users_sigma = self.gen_users_sigma
        p.gen_rpm_mean = self.gen_rpm_mean
        p.gen_window_s = self.gen_window_s
        p.gen_out_edge = self.gen_out_edge
        p.n_edges = self.n_edges
        p.n_servers = self.n_servers
        p.client_out_edge = self.client_out_edge
        p.has_lb = int(self.has_lb)
        p.lb_algo = self.lb_algo
        p.n_lb_edges = len(self.lb_edges)
        p.lb_edges = _ptr(self.lb_edges, C.c_int32)
        p.edge_target_kind = _ptr(self.edge_target_kind, C.c_uint8)
        p.edge_target_idx = _ptr(self.edge_target_idx, C.c_int32)
        p.edge_dist = _ptr(self.edge_dist, C.c_uint8)
        p.edge_mean = _ptr(self.edge_mean, C.c_double)
        p.edge_sigma = _ptr(self.edge_sigma, C.c_double)
        p.edge_dropout = _ptr(self.edge_dropout, C.c_double)
        p.srv_cores = _ptr(self.srv_cores, C.c_uint32)
        p.srv_ram_mb = _ptr(self.srv_ram_mb, C.c_double)
        p.srv_out_edge = _ptr(self.srv_out_edge, C.c_int32)
        p.srv_ep_begin = _ptr(self.srv_ep_begin, C.c_uint32)
        p.n_endpoints = len(self.ep_ram)
        p.ep_step_begin = _ptr(self.ep_step_begin, C.c_uint32)
        p.ep_ram = _ptr(self.ep_ram, C.c_double)
        p.n_steps = len(self.step_kind)
        p.step_kind = _ptr(self.step_kind, C.c_uint8)
        p.step_time = _ptr(self.step_time, C.c_double)
        p.n_edge_marks = len(self.emark_time)
        p.emark_time = _ptr(self.emark_time, C.c_double)
        p.emark_edge = _ptr(self.emark_edge, C.c_int32)
        p.emark_delta = _ptr(self.emark_delta, C.c_double)
        p.n_srv_marks = len(self.smark_time)
        p.smark_time = _ptr(self.smark_time, C.c_double)
        p.smark_lb_edge = _ptr(self.smark_lb_edge, C.c_int32)
        p.smark_down = _ptr(self.smark_down, C.c_uint8)
        p._keepalive = self  # noqa: SLF001 - arrays must outlive the struct
        return p

    # ------------------------------------------------------------------ #
    # sizing helpers                                                      #
    # ------------------------------------------------------------------ #
    def expected_arrivals(self, users_mean: float | None = None, rpm: float | None = None) -> tuple[float, float]:
        """(mean, std) of the number of generated requests over the horizon.

        Compound process of samplers/poisson_poisson.py:20-82: per window W,
        N | U ~ Poisson(U r W), so Var N = E[U] r W + Var(U) (r W)^2.
        """
        u = self.gen_users_mean if users_mean is None else users_mean
        r = (self.gen_rpm_mean if rpm is None else rpm) / 60.0
        w = min(self.gen_window_s, self.total_time)
        n_win = self.total_time / w
        var_u = u if self.gen_users_dist == _abi.DIST_CODES["poisson"] else self.gen_users_sigma**2
        mean = u * r * self.total_time
        var = n_win * (u * r * w + var_u * (r * w) ** 2)
        return mean, math.sqrt(max(var, 0.0))

    def clock_capacity(self, users_mean: float | None = None, rpm: float | None = None) -> int:
        mean, std = self.expected_arrivals(users_mean, rpm)
        return int(mean + 8.0 * std + 64.0)


def _edge_latency_mean(dist: str, m: float, s: float) -> float:
    """Mean transit time of one edge under the reference's variate semantics
    (samplers/common_helpers.py:49-89); only used to size request pools."""
    if dist == "log_normal":
        return math.exp(min(m + 0.5 * s * s, 50.0))
    if dist == "normal":
        return max(m, 0.0) + 0.4 * s
    if dist == "uniform":
        return 0.5
    return m  # exponential, poisson


def lower(payload: Any) -> DevicePlan:  # noqa: C901, PLR0912, PLR0915
    """``SimulationPayload`` | dict  ->  :class:`DevicePlan`.

    Mirrors the wiring of ``SimulationRunner._build_edges``
    (simulation_runner.py:205-260) and the timelines of
    ``EventInjectionRuntime.__init__`` (runtime/events/injection.py:119-163).
    """
    p = normalize_payload(payload)
    rqs, tg, st = p["rqs_input"], p["topology_graph"], p["sim_settings"]
    nodes = tg["nodes"]
    servers, client, lb = nodes["servers"], nodes["client"], nodes["load_balancer"]
    edges = tg["edges"]
    server_ids = [s["id"] for s in servers]
    srv_index = {sid: i for i, sid in enumerate(server_ids)}
    edge_ids = [e["id"] for e in edges]
    edge_index = {eid: i for i, eid in enumerate(edge_ids)}

    # ---- wiring (simulation_runner.py:205-260); later edges win, as the dict
    # assignment `source_object.out_edge = ...` does.
    gen_out, client_out = -1, -1
    srv_out = [-1] * len(servers)
    lb_edges: list[int] = []
    tkind, tidx = [], []
    for i, e in enumerate(edges):
        if e["target"] in srv_index:
            tkind.append(_abi.NODE_SERVER)
            tidx.append(srv_index[e["target"]])
        elif e["target"] == client["id"]:
            tkind.append(_abi.NODE_CLIENT)
            tidx.append(0)
        elif lb is not None and e["target"] == lb["id"]:
            tkind.append(_abi.NODE_LB)
            tidx.append(0)
        else:  # unreachable after validation; the runner raises TypeError here
            msg = f"Unknown runtime for {e['target']!r}"
            raise TypeError(msg)
        src = e["source"]
        if src in srv_index:
            srv_out[srv_index[src]] = i
        elif src == client["id"]:
            client_out = i
        elif src == rqs["id"]:
            gen_out = i
        elif lb is not None and src == lb["id"]:
            lb_edges.append(i)
        else:  # all_nodes[edge.source] -> KeyError in the reference
            msg = f"edge '{e['id']}': unknown source node '{src}'"
            raise ValueError(msg)
    if gen_out < 0:
        msg = "the request generator has no outgoing edge"  # assert at rqs_generator.py:99
        raise ValueError(msg)
    if client_out < 0:
        msg = "the client has no outgoing edge"  # assert at client.py:45
        raise ValueError(msg)
    for i, o in enumerate(srv_out):
        if o < 0:
            msg = f"server '{server_ids[i]}' has no outgoing edge"  # assert at server.py:275
            raise ValueError(msg)
    if lb is not None and not lb_edges:
        msg = "the load balancer has no outgoing edge"
        raise ValueError(msg)

    # ---- servers / endpoints / steps (server.py:95-110: RAM summed up front)
    srv_ep_begin, ep_step_begin, ep_ram = [0], [0], []
    step_kind, step_time = [], []
    step_index: dict[tuple[int, int, int], int] = {}
    for si, s in enumerate(servers):
        if not s["endpoints"]:
            # rng.integers(low=0, high=0) raises ValueError in the reference (server.py:101)
            msg = f"server '{s['id']}' has no endpoints"
            raise ValueError(msg)
        for ei, ep in enumerate(s["endpoints"]):
            ram = 0
            for ki, stp in enumerate(ep["steps"]):
                (op, val), = stp["step_operation"].items()
                if stp["kind"] in RAM_STEP_KINDS:
                    ram = ram + val  # same left-to-right sum as server.py:106-110
                    step_index[(si, ei, ki)] = -1
                elif stp["kind"] in CPU_STEP_KINDS:
                    step_index[(si, ei, ki)] = len(step_kind)
                    step_kind.append(_abi.STEP_CPU)
                    step_time.append(float(val))
                elif stp["kind"] in IO_STEP_KINDS:
                    step_index[(si, ei, ki)] = len(step_kind)
                    step_kind.append(_abi.STEP_IO)
                    step_time.append(float(val))
            ep_ram.append(float(ram))
            ep_step_begin.append(len(step_kind))
        srv_ep_begin.append(len(ep_ram))

    # ---- event timelines (injection.py:119-163)
    emarks: list[tuple[float, int, str, str, float]] = []
    smarks: list[tuple[float, int, str, str]] = []
    edge_by_server: dict[str, int] = {}
    for ei in lb_edges:  # injection.py:158-163: last LB edge per server wins
        edge_by_server[edges[ei]["target"]] = ei
    for ev in p["events"] or []:
        tgt = ev["target_id"]
        if tgt in edge_index:
            spike = float(ev["start"]["spike_s"])
            emarks.append((ev["start"]["t_start"], 1, ev["event_id"], tgt, spike))
            emarks.append((ev["end"]["t_end"], 0, ev["event_id"], tgt, spike))
        elif tgt in srv_index:
            smarks.append((ev["start"]["t_start"], 1, ev["event_id"], tgt))
            smarks.append((ev["end"]["t_end"], 0, ev["event_id"], tgt))
    # sort key (time, mark == start, event_id, target_id): END before START at equal t
    emarks.sort(key=lambda m: (m[0], m[1] == 1, m[2], m[3]))
    smarks.sort(key=lambda m: (m[0], m[1] == 1, m[2], m[3]))

    def _env_times(ts: list[float]) -> list[float]:
        # injection.py:181-188: `dt = t - last_t; if dt > 0: yield timeout(dt)`;
        # the clock after the wait is now + dt, which is what we store.
        out, now, last = [], 0.0, 0.0
        for t in ts:
            dt = t - last
            if dt > 0.0:
                now = now + dt
            last = t
            out.append(now)
        return out

    emark_time = _env_times([m[0] for m in emarks])
    smark_time = _env_times([m[0] for m in smarks])

    # An outage timeline must never leave the load balancer without a live out-edge: the reference
    # then fails inside the run as soon as a request reaches the LB (`next(iter(...))` on an empty
    # OrderedDict -> StopIteration, lb_algorithms.py:33; `min()` of an empty sequence, :19), the
    # payload validators only forbid ALL servers being down at once (schemas/payload.py).  Rejected
    # here, before anything runs.
    if lb is not None and smarks:
        live = list(lb_edges)
        for m in smarks:
            ei = edge_by_server.get(m[3], -1)
            if ei < 0:
                continue
            if ei in live:
                live.remove(ei)
            if m[1] == 0:      # SERVER_UP: re-appended at the tail (injection.py:218-226)
                live.append(ei)
            if not live:
                msg = (f"event '{m[2]}' takes the last live server behind load balancer '{lb['id']}' down at "
                       f"t={m[0]}: the load balancer would have no out-edge to route to")
                raise ValueError(msg)

    mask = 0
    for m in st["enabled_sample_metrics"]:
        mask |= _abi.METRIC_BITS[m]

    users = rqs["avg_active_users"]
    f64 = np.float64
    plan = DevicePlan(
        payload=p,
        total_time=float(st["total_simulation_time"]),
        sample_period=float(st["sample_period_s"]),
        metrics_mask=mask,
        gen_users_dist=_abi.DIST_CODES[users["distribution"]],
        gen_users_mean=float(users["mean"]),
        gen_users_sigma=float(users["variance"] if users["variance"] is not None else 0.0),
        gen_rpm_mean=float(rqs["avg_request_per_minute_per_user"]["mean"]),
        gen_window_s=float(rqs["user_sampling_window"]),
        gen_out_edge=gen_out,
        client_out_edge=client_out,
        has_lb=lb is not None,
        lb_algo=_abi.LB_CODES[lb["algorithms"]] if lb else 0,
        lb_edges=np.asarray(lb_edges, dtype=np.int32).reshape(-1),
        edge_target_kind=np.asarray(tkind, dtype=np.uint8),
        edge_target_idx=np.asarray(tidx, dtype=np.int32),
        edge_dist=np.asarray([_abi.DIST_CODES[e["latency"]["distribution"]] for e in edges], dtype=np.uint8),
        edge_mean=np.asarray([e["latency"]["mean"] for e in edges], dtype=f64),
        edge_sigma=np.asarray(
            [e["latency"]["variance"] if e["latency"]["variance"] is not None else 0.0 for e in edges], dtype=f64
        ),
        edge_dropout=np.asarray([e["dropout_rate"] for e in edges], dtype=f64),
        srv_cores=np.asarray([s["server_resources"]["cpu_cores"] for s in servers], dtype=np.uint32),
        srv_ram_mb=np.asarray([s["server_resources"]["ram_mb"] for s in servers], dtype=f64),
        srv_out_edge=np.asarray(srv_out, dtype=np.int32),
        srv_ep_begin=np.asarray(srv_ep_begin, dtype=np.uint32),
        ep_step_begin=np.asarray(ep_step_begin, dtype=np.uint32),
        ep_ram=np.asarray(ep_ram, dtype=f64),
        step_kind=np.asarray(step_kind, dtype=np.uint8),
        step_time=np.asarray(step_time, dtype=f64),
        emark_time=np.asarray(emark_time, dtype=f64),
        emark_edge=np.asarray([edge_index[m[3]] for m in emarks], dtype=np.int32),
        emark_delta=np.asarray([m[4] if m[1] == 1 else -m[4] for m in emarks], dtype=f64),
        smark_time=np.asarray(smark_time, dtype=f64),
        smark_lb_edge=np.asarray([edge_by_server.get(m[3], -1) for m in smarks], dtype=np.int32),
        smark_down=np.asarray([m[1] for m in smarks], dtype=np.uint8),
        edge_ids=edge_ids,
        server_ids=server_ids,
        step_index=step_index,
    )
    return plan


def estimate_capacities(
    plan: DevicePlan,
    users_max: float | None = None,
    latency_scale: float = 1.0,
    rpm_max: float | None = None,
) -> tuple[int, int]:
    """(request_capacity, fifo_capacity) heuristics for the engine.

    Little's law: live requests ~ arrival rate x time in system, where the time
    in system is the mean transit along generator -> client -> (LB ->) server ->
    client plus the service time, M/D/1 waiting for the CPU, and every injected
    spike.  The number in system is roughly Poisson, so mean + 8 sigma + slack.
    A saturated server (CPU or RAM bound) accumulates a backlog proportional to
    the horizon.  Overflow is detected and reported by the engine, never silent;
    the runner retries with a larger pool.
    """
    payload = plan.payload
    edges = payload["topology_graph"]["edges"]
    users = plan.gen_users_mean if users_max is None else float(users_max)
    rpm = plan.gen_rpm_mean if rpm_max is None else float(rpm_max)
    sd = math.sqrt(max(users, 0.0)) if plan.gen_users_dist == _abi.DIST_CODES["poisson"] else plan.gen_users_sigma
    rate = max(users + 5.0 * sd, 0.0) * rpm / 60.0

    def lat(i: int) -> float:
        e = edges[i]["latency"]
        return _edge_latency_mean(e["distribution"], e["mean"], e["variance"] or 0.0) * latency_scale

    path = lat(plan.gen_out_edge) + lat(plan.client_out_edge)
    if plan.has_lb:
        path += max(lat(int(i)) for i in plan.lb_edges)
    if plan.n_servers:
        path += max(lat(int(i)) for i in plan.srv_out_edge)
    spike = float(plan.emark_delta[plan.emark_delta > 0].sum()) if len(plan.emark_delta) else 0.0

    service = cpu = ram = 0.0
    for ep in range(len(plan.ep_ram)):
        b, e = int(plan.ep_step_begin[ep]), int(plan.ep_step_begin[ep + 1])
        service = max(service, float(plan.step_time[b:e].sum()))
        cpu = max(cpu, float(plan.step_time[b:e][plan.step_kind[b:e] == _abi.STEP_CPU].sum()))
        ram = max(ram, float(plan.ep_ram[ep]))
    n_active = max(1, (len(plan.lb_edges) if plan.has_lb else 1) - (1 if len(plan.smark_time) else 0))
    srv_rate = rate / n_active
    cores = float(plan.srv_cores.min()) if plan.n_servers else 1.0
    rho = srv_rate * cpu / cores
    saturated = rho >= 0.9
    wait = 0.0 if saturated else rho * cpu / (2.0 * (1.0 - rho))
    if ram > 0.0 and plan.n_servers:
        slots = math.floor(float(plan.srv_ram_mb.min()) / ram)
        saturated = saturated or srv_rate * (service + wait) >= 0.8 * slots
    live = rate * (path + spike + service + wait)
    if saturated:
        live += rate * plan.total_time
    cap = int(live + 8.0 * math.sqrt(max(live, 1.0)) + 8.0)
    cap = min(65535, max(16, (cap + 7) // 8 * 8))
    # waiters per server queue: everything when saturated, else the M/D/1 queue length + slack
    lq = 0.0 if saturated else rho * rho / (2.0 * (1.0 - rho))
    want = cap if saturated else min(cap, int(lq + 10.0 * math.sqrt(lq + 1.0)) + 1)
    fifo = 8
    while fifo < want:
        fifo *= 2
    return cap, fifo
