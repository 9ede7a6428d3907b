"""``SimulationRunner``: host-side mirror of the reference's entry point.

Reference seam (the ONLY boundary this path has):

    SimulationRunner(env=simpy.Environment(), simulation_input=payload).run() -> ResultsAnalyzer
    SimulationRunner.from_yaml(env=..., yaml_path=...)
    (/root/reference/src/asyncflow/runtime/simulation_runner.py:52-57, 349-398)

Same constructor keywords, same ``run()`` / ``from_yaml`` names.  Differences,
all additive: ``env`` is accepted and ignored (there is no SimPy environment:
the HIP kernel owns the clock); ``replicas`` / ``seeds`` / ``sweep`` describe
the batch; ``run()`` returns :class:`BatchedResults`, whose items expose the
reference analyzer's accessors.  With the defaults (1 replica) ``run()`` returns
a single :class:`ScenarioResults`, i.e. a drop-in for the reference call.
"""

from __future__ import annotations

import re
import time
import warnings
from pathlib import Path
from typing import Any, Mapping, Sequence

import numpy as np

from . import _abi
from .engine import Engine
from .payload import load_yaml, normalize_payload
from .plan import DevicePlan, estimate_capacities, lower
from .results import LATENCY_KEYS, BatchedResults, ScenarioResults

DEFAULT_SEED_BASE = 0x5EED0000  # BASELINE config 2: scenario i uses Philox key 0x5EED0000 + i

_EDGE_RE = re.compile(r"^topology_graph\.edges\[(?P<id>[^\]]+)\]\.(?P<f>latency\.mean|latency\.variance|dropout_rate)$")
_STEP_RE = re.compile(
    r"^topology_graph\.nodes\.servers\[(?P<sid>[^\]]+)\]\.endpoints\[(?P<ep>\d+)\]\.steps\[(?P<k>\d+)\]\.(?P<f>cpu_time|io_waiting_time)$"
)
_RES_RE = re.compile(r"^topology_graph\.nodes\.servers\[(?P<sid>[^\]]+)\]\.server_resources\.(?P<f>cpu_cores|ram_mb)$")
_EVENT_RE = re.compile(r"^events\[(?P<id>[^\]]+)\]\.(?P<f>start\.t_start|end\.t_end|start\.spike_s)$")


#: payloads (by content hash) whose sweeps met instants shared by several timed events
_SHARED_INSTANTS_SEEN: dict[str, bool] = {}


def _as_int(key: str, value: float) -> int:
    if float(value) != int(value):
        msg = f"sweep key {key!r}: {value!r} is not an integer"
        raise ValueError(msg)
    return int(value)


def write_point(payload: dict, key: str, value: float) -> None:
    """Write ONE sweep value into a (deep-copied, normalised) payload dict at its YAML-style path: the payload a user
    of the reference would have built for that grid point.  Used for per-point validation through the payload models
    and to lower the event timelines of each distinct combination of event parameters."""
    value = float(value)
    if key.startswith("rqs_input.avg_active_users."):
        payload["rqs_input"]["avg_active_users"][key.rsplit(".", 1)[1]] = value
    elif key == "rqs_input.avg_request_per_minute_per_user.mean":
        payload["rqs_input"]["avg_request_per_minute_per_user"]["mean"] = value
    elif key == "rqs_input.user_sampling_window":
        payload["rqs_input"]["user_sampling_window"] = _as_int(key, value)       # an int field (rqs_generator.py:17-27)
    elif m := _EDGE_RE.match(key):
        hit = False
        for e in payload["topology_graph"]["edges"]:
            if m["id"] in ("*", e["id"]):
                hit = True
                if m["f"] == "dropout_rate":
                    e["dropout_rate"] = value
                else:
                    e["latency"][m["f"].split(".")[1]] = value
        if not hit:
            msg = f"sweep key {key!r}: unknown edge id {m['id']!r}"
            raise ValueError(msg)
    elif m := _STEP_RE.match(key):
        srv = next((s for s in payload["topology_graph"]["nodes"]["servers"] if s["id"] == m["sid"]), None)
        if srv is None:
            msg = f"sweep key {key!r}: unknown server id"
            raise ValueError(msg)
        try:
            step = srv["endpoints"][int(m["ep"])]["steps"][int(m["k"])]
        except IndexError:
            msg = f"sweep key {key!r}: no such step"
            raise ValueError(msg) from None
        if m["f"] not in step["step_operation"]:
            msg = f"sweep key {key!r}: not a CPU or I/O step"
            raise ValueError(msg)
        step["step_operation"] = {m["f"]: value}
    elif m := _RES_RE.match(key):
        srv = next((s for s in payload["topology_graph"]["nodes"]["servers"] if s["id"] == m["sid"]), None)
        if srv is None:
            msg = f"sweep key {key!r}: unknown server id"
            raise ValueError(msg)
        srv["server_resources"][m["f"]] = _as_int(key, value)                  # PositiveInt fields (nodes.py:58-69)
    elif m := _EVENT_RE.match(key):
        ev = next((v for v in (payload.get("events") or []) if v["event_id"] == m["id"]), None)
        if ev is None:
            msg = f"sweep key {key!r}: unknown event id {m['id']!r}"
            raise ValueError(msg)
        part, field = m["f"].split(".")
        if field == "spike_s" and ev["start"].get("spike_s") is None:
            msg = f"sweep key {key!r}: the event is not a network spike"
            raise ValueError(msg)
        ev[part][field] = value
    else:
        msg = f"unsupported sweep key {key!r}"
        raise ValueError(msg)


VALIDATE_EVERY_POINT_UP_TO = 512      # distinct points validated row by row; the reference's model costs ~0.2 ms per payload
MAX_ATTEMPTS = 6   # runs of one sweep while the engine reports a capacity overflow (SimulationRunner.run: how the pools grow)
_VALIDATED: dict[tuple, int] = {}       # (plan, columns) digests already validated in this process


def validate_points(plan: DevicePlan, columns: Mapping[str, np.ndarray], n: int) -> int:
    """Per-point validation of a sweep through the payload models -- the reference's own
    ``SimulationPayload.model_validate`` when ``asyncflow`` is importable (payload.py:20-252 validates every payload it
    runs), else the structural validators of asyncflow_amd/payload.py (``normalize_payload`` picks).

    * Up to ``VALIDATE_EVERY_POINT_UP_TO`` DISTINCT points (rows of the column table; seed replicas of one point count
      once): every one of them is validated, like the reference would.
    * Larger sweeps (a 100 x 100 grid is 10 000 points = seconds of model validation per call, and `bench.py` resolves
      a sweep several times): every DISTINCT VALUE of every column is validated inside the first row of the
      sweep that holds it (the schema's field constraints are per field; its cross-field constraints see the row's own
      values), and so are the rows that hold every column's
      extremes, the first and the last one, and 64 rows spread evenly over the sweep -- the schema's cross-field
      constraints (a Gaussian's variance with its mean, the sampling window with the horizon) are monotone in each
      column, the constraints that are not -- integrality of the int fields -- are checked over whole columns here and in
      `resolve_sweep`, and cross-column constraints of the event columns are covered exhaustively by `_event_columns`
      (every distinct combination is lowered).  The reduced coverage is logged (logger ``asyncflow_amd``).
    * A (plan, columns) pair validated once in this process is not validated again.
    Returns the number of payloads validated; raises ``ValueError`` (pydantic's ValidationError is one)."""
    import copy
    import hashlib
    import json
    import logging

    from .payload import normalize_payload

    keys = list(columns)
    for key in keys:                      # integrality is not monotone: whole column, whatever the number of points
        if key == "rqs_input.user_sampling_window" or _RES_RE.match(key):
            col = columns[key]
            if np.any(col != np.floor(col)):
                bad = int(np.argmax(col != np.floor(col)))
                msg = f"sweep key {key!r}: an integer field of the schema, but scenario {bad} has {float(col[bad])!r}"
                raise ValueError(msg)
    table = np.stack([np.asarray(columns[k], dtype=np.float64) for k in keys], axis=1) if keys else np.zeros((n, 0))
    digest = hashlib.sha1(json.dumps(plan.payload, sort_keys=True, default=str).encode())
    digest.update(repr(keys).encode())
    digest.update(np.ascontiguousarray(table).tobytes())
    memo = (digest.hexdigest(), n)
    if memo in _VALIDATED:
        return _VALIDATED[memo]

    def check(values: Mapping[str, float], what: str) -> None:
        point = copy.deepcopy(plan.payload)
        for key, value in values.items():
            write_point(point, key, value)
        try:
            normalize_payload(point)
        except ValueError as exc:
            msg = f"{what} ({ {k: float(v) for k, v in values.items()} }) is not a valid payload: {exc}"
            raise ValueError(msg) from exc

    _, first = np.unique(table, axis=0, return_index=True)
    done = 0
    if len(first) <= VALIDATE_EVERY_POINT_UP_TO:
        picks = sorted(int(i) for i in first)
    else:
        # every distinct value of every column -- inside a row of the sweep that holds it (the first one), not on its own in
        # the plan's base payload: the schema's cross-field constraints (an event's t_start < t_end, an event inside the
        # horizon, a Gaussian's variance with its mean) relate a row's OWN values, and a value that is valid in its rows can
        # be invalid beside the base payload's (t_start = 160.07 against the base t_end = 160)
        rows: set[int] = set()
        for key in keys:
            _, at = np.unique(columns[key], return_index=True)
            rows.update(int(i) for i in at)
        done = 0
        rows |= {0, n - 1} | {int(i) for i in np.linspace(0, n - 1, 64).round()}
        for col in columns.values():
            rows.add(int(np.argmin(col)))
            rows.add(int(np.argmax(col)))
        picks = sorted(rows)
        logging.getLogger("asyncflow_amd").info(
            "sweep of %d distinct points: validated %d whole rows (the first row holding every distinct value of every column, "
            "the columns' extremes, both ends, 64 spread evenly) instead of every row", len(first), len(picks))
    for i in picks:
        check({key: col[i] for key, col in columns.items()}, f"sweep point {i}")
    _VALIDATED[memo] = done + len(picks)
    if len(_VALIDATED) > 64:
        _VALIDATED.pop(next(iter(_VALIDATED)))
    return done + len(picks)


def resolve_sweep(plan: DevicePlan, sweep: Mapping[str, Any] | None, n: int) -> list[tuple[int, int, np.ndarray, str]]:
    """Map YAML-style paths to ``af_override_t`` columns.

    Accepted keys (values: one float per scenario, or one for all):
      rqs_input.avg_active_users.mean | .variance
      rqs_input.avg_request_per_minute_per_user.mean
      rqs_input.user_sampling_window
      topology_graph.edges[<edge id>|*].latency.mean | .latency.variance | .dropout_rate
      topology_graph.nodes.servers[<id>].endpoints[<j>].steps[<k>].cpu_time | .io_waiting_time
      topology_graph.nodes.servers[<id>].server_resources.cpu_cores | .ram_mb
      events[<event id>].start.t_start | .end.t_end | .start.spike_s

    Every point is validated by the payload models (`validate_points`).  Event columns do not map one to one to engine
    columns: the reference sorts the marks of all events into two timelines (injection.py:142-151) and re-accumulates
    the waits between them (:181-188), so each DISTINCT combination of event values is lowered on its own (`lower()`:
    validation + timelines) and every mark slot of the plan gets its (time, delta, edge) / (time, LB edge, down)
    columns -- scenarios whose marks sort differently are exact too.
    """
    out: list[tuple[int, int, np.ndarray, str]] = []
    cols = {key: np.ascontiguousarray(np.broadcast_to(np.asarray(values, dtype=np.float64), (n,))) for key, values in (sweep or {}).items()}
    code = _abi.PARAM_CODES
    event_cols: dict[str, np.ndarray] = {}
    for key, col in cols.items():
        if key == "rqs_input.avg_active_users.mean":
            out.append((code["gen_users_mean"], 0, col, key))
        elif key == "rqs_input.avg_active_users.variance":
            out.append((code["gen_users_sigma"], 0, col, key))
        elif key == "rqs_input.avg_request_per_minute_per_user.mean":
            out.append((code["gen_rpm_mean"], 0, col, key))
        elif key == "rqs_input.user_sampling_window":
            out.append((code["gen_window"], 0, col, key))
        elif m := _EDGE_RE.match(key):
            name = {"latency.mean": "edge_mean", "latency.variance": "edge_sigma", "dropout_rate": "edge_dropout"}[m["f"]]
            ids = plan.edge_ids if m["id"] == "*" else [m["id"]]
            for eid in ids:
                if eid not in plan.edge_ids:
                    msg = f"sweep key {key!r}: unknown edge id {eid!r}"
                    raise ValueError(msg)
                out.append((code[name], plan.edge_ids.index(eid), col, key))
            if name == "edge_mean" and np.any(col <= 0):
                msg = f"sweep key {key!r}: edge latency mean must be positive"  # edges.py:79-81
                raise ValueError(msg)
        elif m := _STEP_RE.match(key):
            if m["sid"] not in plan.server_ids:
                msg = f"sweep key {key!r}: unknown server id"
                raise ValueError(msg)
            idx = plan.step_index.get((plan.server_ids.index(m["sid"]), int(m["ep"]), int(m["k"])), -1)
            if idx < 0:
                msg = f"sweep key {key!r}: not a CPU or I/O step"
                raise ValueError(msg)
            if np.any(col <= 0):
                msg = f"sweep key {key!r}: step times must be positive"   # PositiveFloat in the reference schema
                raise ValueError(msg)
            out.append((code["step_time"], idx, col, key))
        elif m := _RES_RE.match(key):
            if m["sid"] not in plan.server_ids:
                msg = f"sweep key {key!r}: unknown server id"
                raise ValueError(msg)
            if np.any(col != np.floor(col)) or np.any(col < 1):
                msg = f"sweep key {key!r}: positive integers expected"     # PositiveInt (nodes.py:58-69)
                raise ValueError(msg)
            out.append((code["srv_cores" if m["f"] == "cpu_cores" else "srv_ram_mb"], plan.server_ids.index(m["sid"]), col, key))
        elif _EVENT_RE.match(key):
            event_cols[key] = col
        else:
            msg = f"unsupported sweep key {key!r}"
            raise ValueError(msg)
    if cols:
        validate_points(plan, cols, n)
    if event_cols:
        out.extend(_event_columns(plan, event_cols, n))
    return out


def _event_columns(plan: DevicePlan, event_cols: Mapping[str, np.ndarray], n: int) -> list[tuple[int, int, np.ndarray, str]]:
    """Timeline columns of a sweep over event parameters: one `lower()` per distinct combination of event values."""
    import copy

    keys = list(event_cols)
    table = np.stack([event_cols[k] for k in keys], axis=1)                     # [n, k]
    uniq, inverse = np.unique(table, axis=0, return_inverse=True)
    inverse = np.asarray(inverse).reshape(-1)
    if len(uniq) > 20_000:
        msg = f"{len(uniq)} distinct combinations of event parameters: each is validated and lowered on its own (limit 20 000)"
        raise ValueError(msg)
    n_em, n_sm = len(plan.emark_time), len(plan.smark_time)
    em = np.zeros((len(uniq), 3, n_em))
    sm = np.zeros((len(uniq), 3, n_sm))
    for u, row in enumerate(uniq):
        point = copy.deepcopy(plan.payload)
        for key, value in zip(keys, row):
            write_point(point, key, value)
        try:
            lp = lower(point)                 # validates (payload models) and builds the two sorted timelines
        except ValueError as exc:
            msg = f"event sweep point { {k: float(v) for k, v in zip(keys, row)} } is not a valid payload: {exc}"
            raise ValueError(msg) from exc
        if len(lp.emark_time) != n_em or len(lp.smark_time) != n_sm:
            msg = "an event sweep must not change the number of timeline marks"
            raise ValueError(msg)
        em[u] = np.stack([lp.emark_time, lp.emark_delta, lp.emark_edge.astype(np.float64)]) if n_em else em[u]
        sm[u] = np.stack([lp.smark_time, lp.smark_lb_edge.astype(np.float64), lp.smark_down.astype(np.float64)]) if n_sm else sm[u]
    out: list[tuple[int, int, np.ndarray, str]] = []
    code = _abi.PARAM_CODES
    label = "events[...]: " + ", ".join(keys)
    for i in range(n_em):     # (a slot that equals the plan's in every scenario needs no column)
        for j, name in enumerate(("emark_time", "emark_delta", "emark_edge")):
            colv = np.ascontiguousarray(em[inverse, j, i])
            if np.any(colv != (plan.emark_time, plan.emark_delta, plan.emark_edge)[j][i]):
                out.append((code[name], i, colv, f"{label} -> {name}[{i}]"))
    for i in range(n_sm):
        for j, name in enumerate(("smark_time", "smark_lb_edge", "smark_down")):
            colv = np.ascontiguousarray(sm[inverse, j, i])
            if np.any(colv != (plan.smark_time, plan.smark_lb_edge, plan.smark_down)[j][i]):
                out.append((code[name], i, colv, f"{label} -> {name}[{i}]"))
    return out


def _fifo_pow2(want: int) -> int:
    """Smallest power of two >= ``want``, clamped to the engine's wait-queue limit."""
    p = 8
    while p < want and p < _abi.MAX_FIFO_CAPACITY:
        p *= 2
    return p


class SimulationRunner:
    """Build -> lower -> run the batched HIP engine -> results."""

    def __init__(
        self,
        *,
        env: Any = None,
        simulation_input: Any,
        replicas: int = 1,
        seeds: Sequence[int] | np.ndarray | None = None,
        sweep: Mapping[str, Any] | None = None,
        device: int | None = None,
        request_capacity: int | None = None,
        fifo_capacity: int | None = None,
        clock_capacity: int | None = None,
        collect_clock: bool = True,
        collect_samples: bool = True,
        force_global_state: bool = False,
        auto_grow: bool = True,
        lanes_per_wave: int = 0,
        draw_memory_mb: int = 0,
        expect_shared_instants: bool | None = None,
        specialise: bool | None = None,
        online_summary: Mapping[str, Any] | None = None,
        flow: bool | str = True,
        flow_list_entries: int = 0,
        flow_ring_rows: int = 0,
        devices: Sequence[int] | None = None,
        on_negative_delay: str = "raise",
        summary: Mapping[str, Any] | bool | None = None,
    ) -> None:
        self.env = env  # accepted for signature compatibility; unused
        #: the keyword arguments of `BatchedResults.summary()` (True: its defaults): run() then makes ONE engine call for simulation
        #: and analyzer (`af_engine_run_summarized`: the analyzer of the stage-parallel kernel's full residency rounds runs beside
        #: its last, partial one) and `results.summary(**the same arguments)` returns what that call wrote
        self.summary_kw: dict[str, Any] | None = ({} if summary is True else dict(summary)) if summary else None
        self.simulation_input = simulation_input
        self.payload = normalize_payload(simulation_input)
        self.plan: DevicePlan = lower(self.payload)
        if seeds is not None:
            self.seeds = np.asarray(seeds, dtype=np.uint64).reshape(-1)
        else:
            self.seeds = (DEFAULT_SEED_BASE + np.arange(int(replicas), dtype=np.uint64)).astype(np.uint64)
        if self.seeds.size == 0:
            msg = "at least one scenario is required"
            raise ValueError(msg)
        self.sweep = dict(sweep or {})
        self.device = device
        self.request_capacity = request_capacity
        self.fifo_capacity = fifo_capacity
        self.clock_capacity = clock_capacity
        self.collect_clock = collect_clock
        self.collect_samples = collect_samples
        #: {"hist_max": seconds, "hist_bins": 4096}: let the kernel itself accumulate a latency histogram
        #: and the 1-s completion counts per scenario (for sweeps run with collect_clock=False)
        self.online_summary = dict(online_summary) if online_summary else None
        if self.online_summary is not None and not float(self.online_summary.get("hist_max", 0.0)) > 0.0:
            msg = "online_summary needs a positive 'hist_max' (seconds covered by the latency histogram)"
            raise ValueError(msg)
        self.force_global_state = force_global_state
        self.auto_grow = auto_grow
        self.lanes_per_wave = lanes_per_wave
        self.draw_memory_mb = draw_memory_mb
        #: True: start with the kernel variant that replays SimPy's event order at instants shared by
        #: several timed events; None: start lean, hand such scenarios over, and remember (per
        #: payload, in this process) that this topology produces them
        self.expect_shared_instants = expect_shared_instants
        #: plan-specialised kernels (asyncflow_amd/jit.py, ~3 s of hipcc once per plan shape, cached on disk):
        #: True = build them, False = the library's generic kernels, None = take them from the cache when an earlier
        #: sweep left them there and build only when the sweep is long enough to repay it (> 1e9 request-events on the
        #: next-event kernels, > 5e10 on the stage-parallel kernel)
        self.specialise = specialise
        #: stage-parallel kernel (one wave per scenario, 64 requests per step) for plans in its range
        #: (Engine.flow_reason()); False = always the next-event kernels.  Results are bit-identical.  True also lets a
        #: sweep the kernel cannot be sized for (a cpu_cores column above 64) fall back to the next-event kernels;
        #: "always" makes that an error.
        self.flow = flow
        self.flow_list_entries = flow_list_entries
        self.flow_ring_rows = flow_ring_rows
        #: several GPUs from ONE process: scenarios are dealt to the devices (by expected load when the sweep
        #: has a users column; contiguous otherwise), one engine per device runs in its own host
        #: thread, no exchange during simulation.  (One process per GPU + the RCCL gather: bench.py --gpus N.)
        self.devices = [int(d) for d in devices] if devices else None
        #: "raise" (the reference's behaviour: ValueError "Negative delay", edge.py:107 / simpy) or "flag" (keep the results,
        #: the scenario carries AF_FLAG_NEGATIVE_DELAY) for a send whose transit + spike is negative
        if on_negative_delay not in ("raise", "flag"):
            msg = "on_negative_delay must be 'raise' or 'flag'"
            raise ValueError(msg)
        self.on_negative_delay = on_negative_delay
        self._init_kwargs = dict(replicas=replicas, device=device, request_capacity=request_capacity,
                                 fifo_capacity=fifo_capacity, clock_capacity=clock_capacity, collect_clock=collect_clock,
                                 collect_samples=collect_samples, force_global_state=force_global_state, auto_grow=auto_grow,
                                 lanes_per_wave=lanes_per_wave, draw_memory_mb=draw_memory_mb,
                                 expect_shared_instants=expect_shared_instants, specialise=specialise,
                                 online_summary=online_summary, flow=flow, flow_list_entries=flow_list_entries, summary=summary,
                                 flow_ring_rows=flow_ring_rows, on_negative_delay=on_negative_delay)
        self._single = seeds is None and int(replicas) == 1 and not self.sweep
        self._engine: Engine | None = None

    # ------------------------------------------------------------------ #
    def _capacities(self, overrides: list[tuple[int, int, np.ndarray, str]]) -> tuple[int, int, int]:
        users_max, rpm_max, lat_scale = None, None, 1.0
        for code, _idx, col, _ in overrides:
            if code == _abi.PARAM_CODES["gen_users_mean"]:
                users_max = float(col.max())
            elif code == _abi.PARAM_CODES["gen_rpm_mean"]:
                rpm_max = float(col.max())
            elif code == _abi.PARAM_CODES["edge_mean"]:
                base = float(self.plan.edge_mean[_idx]) or 1.0
                lat_scale = max(lat_scale, float(col.max()) / base)
        cap, fifo = estimate_capacities(self.plan, users_max, lat_scale, rpm_max)
        cap = int(self.request_capacity or min(cap, _abi.MAX_REQUEST_CAPACITY))
        # (the engine rounds the FIFO capacity up to a power of two and refuses more than MAX_FIFO_CAPACITY)
        fifo = int(self.fifo_capacity or _fifo_pow2(min(fifo, cap)))
        clock_cap = int(self.clock_capacity or self.plan.clock_capacity(users_max, rpm_max))
        return cap, fifo, clock_cap

    def _want_specialised(self, n: int, clock_cap: int, on_flow_kernel: bool = False) -> bool:
        # ~7 request-events per completed request; clock_cap bounds the completions of one scenario.  A hipcc run (~3 s) pays
        # where it saves more: the stage-parallel kernel gains ~0.05 ns per request-event in its tandem form (64 -> 43 ms per
        # 5.5e9) and in its general-server form alike (1 360 -> 1 052 ms per 6.2e9, measured round 4), the next-event kernels
        # 0.04 ns but they run 50 x longer per event
        return 7.0 * clock_cap * n > (5e10 if on_flow_kernel else 1e9)

    def _plan_key(self) -> str:
        import hashlib
        import json

        return hashlib.sha1(json.dumps(self.plan.payload, sort_keys=True, default=str).encode()).hexdigest()

    def _run_sharded(self) -> Any:
        import threading

        from .distributed import interleave_by_load, shard_bounds
        from .results import ShardedResults

        n, k = int(self.seeds.size), len(self.devices)
        users = self.sweep.get("rqs_input.avg_active_users.mean")
        if users is not None:
            index = interleave_by_load(np.broadcast_to(np.asarray(users, dtype=np.float64), (n,)), k)
        else:
            index = [np.arange(*shard_bounds(n, r, k)) for r in range(k)]
        shards: list[Any] = [None] * k
        errors: list[BaseException] = []

        def work(r: int) -> None:
            try:
                cols = {key: np.ascontiguousarray(np.broadcast_to(np.asarray(v, dtype=np.float64), (n,))[index[r]])
                        for key, v in self.sweep.items()}
                kw = dict(self._init_kwargs, device=self.devices[r])
                kw.pop("replicas")
                child = SimulationRunner(simulation_input=self.simulation_input, seeds=self.seeds[index[r]], sweep=cols, **kw)
                shards[r] = child.run()
            except BaseException as exc:  # noqa: BLE001 - re-raised in the caller's thread
                errors.append(exc)

        t0 = time.perf_counter()
        threads = [threading.Thread(target=work, args=(r,)) for r in range(k) if len(index[r])]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        keep = [r for r in range(k) if shards[r] is not None]
        return ShardedResults([shards[r] for r in keep], [index[r] for r in keep], time.perf_counter() - t0)

    def _engine_kw(self, cap: int, fifo: int) -> dict[str, Any]:
        return dict(request_capacity=cap, fifo_capacity=fifo, force_global_state=self.force_global_state,
                    lanes_per_wave=self.lanes_per_wave, draw_memory_mb=self.draw_memory_mb, flow=self.flow,
                    flow_list_entries=self.flow_list_entries, flow_ring_rows=self.flow_ring_rows,
                    expect_shared_instants=(_SHARED_INSTANTS_SEEN.get(self._plan_key(), False)
                                            if self.expect_shared_instants is None else bool(self.expect_shared_instants)))

    def jit_spec(self) -> str:
        """The ``-D`` flags of the plan-specialised kernels the FIRST run of this sweep launches (`asyncflow_amd/jit.py`), from a
        planning-only engine: no GPU needed.  Of the output buffers only their presence and capacities enter, so the shape is
        that of `run()` -- the same capacities estimate, clock / samples / kernel-side summary as this runner was set up.  Sweeps
        the next-event kernels run (`flow=False`, plans outside the stage-parallel range) raise `EngineUnavailableError`: the
        shape of those kernels depends on the device."""
        from .engine import PLAN_ONLY

        n = int(self.seeds.size)
        overrides = resolve_sweep(self.plan, self.sweep, n)
        cap, fifo, clock_cap = self._capacities(overrides)
        o_bins = int(self.online_summary.get("hist_bins", 4096)) if self.online_summary is not None else 0
        o_buckets = int(self.plan.total_time) if self.online_summary is not None else 0
        eng = Engine(self.plan, PLAN_ONLY, **self._engine_kw(cap, fifo))
        try:
            return eng.jit_spec(self.seeds, [(c, i, v) for c, i, v, _ in overrides],
                                clock_ptr=8 if self.collect_clock else 0, clock_capacity=clock_cap,
                                samples_ptr=8 if self.collect_samples else 0, tick_capacity=max(self.plan.tick_count, 1), counts_ptr=8,
                                draw_capacity=clock_cap, online_hist_ptr=8 if o_bins else 0, online_hist_bins=o_bins,
                                online_hist_max=float(self.online_summary["hist_max"]) if o_bins else 0.0,
                                online_rps_ptr=8 if o_buckets else 0, online_rps_buckets=o_buckets)
        finally:
            eng.close()

    def prebuild(self) -> str:
        """Build (or find in the cache) the plan-specialised kernel of this sweep WITHOUT a GPU: run it where hipcc is -- a build
        machine, a container image step -- and ship `asyncflow_amd/csrc/_jit/` with the package; `run()` on a box without a
        compiler then loads the specialised kernel instead of the library's generic one.  Returns the spec."""
        from . import jit

        spec = self.jit_spec()
        if spec:
            jit.code_object(spec)
        return spec

    @staticmethod
    def prebuild_reference_examples(verbose: bool = False) -> dict[str, str]:
        """`prebuild()` for each of the reference's own example inputs (`workloads.reference_examples()`).  The kernel a sweep
        launches does not depend on its replica count, so ONE build per example serves every seed-replica sweep of it: the
        examples a user of the reference starts from run on their plan-specialised kernel out of the box (64 -> 38 ms per
        10 000 replicas of `two_servers_lb.yml`), a payload of the user's own from its second sweep on (`jit.build_in_background`).
        Called by `__graft_entry__.build()`; needs hipcc, not a GPU.  Returns {example: spec}."""
        from . import workloads

        specs: dict[str, str] = {}
        for name, payload in workloads.reference_examples().items():
            t0 = time.perf_counter()
            specs[name] = SimulationRunner(simulation_input=payload, replicas=16).prebuild()
            if verbose:
                print(f"prebuilt the kernel of {name} in {time.perf_counter() - t0:.1f} s", flush=True)
        return specs

    def run(self) -> BatchedResults | ScenarioResults:
        """Lower once, launch the HIP kernel over every scenario, return results."""
        import torch

        if self.devices is not None:
            if self.device is not None and int(self.device) not in self.devices:
                msg = f"device={self.device} is not one of devices={self.devices}: pass one or the other"
                raise ValueError(msg)
            if len(self.devices) > 1 and int(self.seeds.size) > 1:
                return self._run_sharded()
            if self.device is None:          # one scenario (or one device): nothing to deal out
                self.device = self.devices[0]

        if not torch.cuda.is_available():
            from .engine import EngineUnavailableError

            msg = "no GPU visible (torch.cuda.is_available() is False): asyncflow_amd has no CPU fallback"
            raise EngineUnavailableError(msg)
        device = torch.cuda.current_device() if self.device is None else int(self.device)
        n = int(self.seeds.size)
        overrides = resolve_sweep(self.plan, self.sweep, n)
        cap, fifo, clock_cap = self._capacities(overrides)
        dev = torch.device("cuda", device)
        ticks = max(self.plan.tick_count, 1)
        t0 = time.perf_counter()
        pool_overflows = 0
        for attempt in range(MAX_ATTEMPTS):
            eng = Engine(self.plan, device, **self._engine_kw(cap, fifo))
            self._engine = eng
            counts = torch.zeros((n, _abi.CNT_SLOTS), dtype=torch.int32, device=dev)
            clock = torch.empty((n, clock_cap, 2), dtype=torch.float64, device=dev) if self.collect_clock else None
            samples = (
                torch.zeros((n, ticks, self.plan.series_pitch), dtype=torch.int32, device=dev)
                if self.collect_samples else None
            )
            online_hist = online_rps = None
            o_bins = o_buckets = 0
            o_max = 0.0
            if self.online_summary is not None:
                o_bins = int(self.online_summary.get("hist_bins", 4096))
                o_max = float(self.online_summary["hist_max"])
                o_buckets = int(self.plan.total_time)
                online_hist = torch.zeros((n, o_bins), dtype=torch.int32, device=dev)
                online_rps = torch.zeros((n, max(o_buckets, 1)), dtype=torch.int32, device=dev)
            torch.cuda.synchronize(dev)
            run_kw = dict(clock_ptr=clock.data_ptr() if clock is not None else 0, clock_capacity=clock_cap,
                          samples_ptr=samples.data_ptr() if samples is not None else 0, tick_capacity=ticks,
                          counts_ptr=counts.data_ptr(), draw_capacity=clock_cap)
            cols = [(c, i, v) for c, i, v, _ in overrides]
            run_kw.update(online_hist_ptr=online_hist.data_ptr() if online_hist is not None else 0, online_hist_bins=o_bins,
                          online_hist_max=o_max,
                          online_rps_ptr=online_rps.data_ptr() if online_rps is not None and o_buckets else 0,
                          online_rps_buckets=o_buckets)
            summ_out: dict[str, Any] | None = None
            summ_ptrs: dict[str, Any] | None = None
            if self.summary_kw is not None and clock is not None:
                kw = {"rps": True, "hist_bins": 0, "hist_max": 0.0, "series": False, **self.summary_kw}
                if not (kw["series"] and samples is None):
                    T_s = int(self.plan.total_time)
                    summ_out = {"stats": torch.empty((n, 8), dtype=torch.float64, device=dev), "keys": LATENCY_KEYS}
                    if kw["rps"] and T_s > 0:
                        summ_out["rps"] = torch.empty((n, T_s), dtype=torch.float32, device=dev)
                    if kw["hist_bins"]:
                        summ_out["hist"] = torch.empty((n, int(kw["hist_bins"])), dtype=torch.int32, device=dev)
                    if kw["series"]:
                        summ_out["series_mean"] = torch.empty((n, self.plan.n_series), dtype=torch.float64, device=dev)
                        summ_out["series_max"] = torch.empty((n, self.plan.n_series), dtype=torch.int32, device=dev)
                    summ_ptrs = dict(stats_ptr=summ_out["stats"].data_ptr(),
                                     rps_ptr=summ_out["rps"].data_ptr() if "rps" in summ_out else 0, rps_buckets=T_s if "rps" in summ_out else 0,
                                     hist_ptr=summ_out["hist"].data_ptr() if "hist" in summ_out else 0, hist_bins=int(kw["hist_bins"]),
                                     hist_max=float(kw["hist_max"]),
                                     series_mean_ptr=summ_out["series_mean"].data_ptr() if kw["series"] else 0,
                                     series_max_ptr=summ_out["series_max"].data_ptr() if kw["series"] else 0)
                    summ_out["_kw"] = kw
            torch.cuda.synchronize(dev)
            build = True
            if self.specialise is None:
                # which kernel family this sweep runs on is the ENGINE's decision (plan range, a sweep it cannot be sized
                # for): read it off the spec it would launch -- asked with the very output shape of the launch, kernel-side
                # summary included, so that the probe plans the instantiation that runs
                on_flow_kernel = "-DAF_FLOW_JIT=1" in eng.jit_spec(self.seeds, cols, **run_kw)
                build = self._want_specialised(n, clock_cap, on_flow_kernel)
            stats = eng.run(
                self.seeds,
                cols,
                **run_kw,
                specialise=True if self.specialise is None else bool(self.specialise),
                specialise_build=build,
                summary=summ_ptrs,
            )
            flow_reason = eng.flow_reason() if self.flow else "flow=False"
            if self.flow and not flow_reason and int(stats.flow_scenarios) == 0:
                flow_reason = "the stage-parallel kernel could not be sized for this sweep (flow='always' reports why)"
            eng.close()
            if int(stats.shared_instant_scenarios) > 0:
                _SHARED_INSTANTS_SEEN[self._plan_key()] = True
            user_cols = {k: np.ascontiguousarray(np.broadcast_to(np.asarray(v, dtype=np.float64), (n,))) for k, v in self.sweep.items()}
            res = BatchedResults(self.plan, self.seeds, counts, clock, samples, stats,
                                 time.perf_counter() - t0, user_cols,
                                 online_hist=online_hist, online_rps=online_rps, online_hist_max=o_max)
            res.flow_reason = flow_reason
            if summ_out is not None:
                summ_out["summary_ms"] = float(stats.summary_ms)
                res._summary_from_run = summ_out  # noqa: SLF001
            over = int(np.bitwise_or.reduce(res.flags)) & _abi.FATAL_FLAGS
            if not over or attempt == MAX_ATTEMPTS - 1 or not self.auto_grow:
                break
            # capacities are estimates (Little's law); overflow is flagged by the
            # kernel, never silent -> grow the overflowing pool and run again.  A pool that overflowed at its real
            # maximum cannot be helped by another run: raise_on_overflow() below reports it.
            if (over & _abi.FLAG_POOL_OVERFLOW and cap >= _abi.MAX_REQUEST_CAPACITY) or \
                    (over & _abi.FLAG_FIFO_OVERFLOW and fifo >= _abi.MAX_FIFO_CAPACITY):
                break
            if over & (_abi.FLAG_CLOCK_OVERFLOW | _abi.FLAG_DRAW_OVERFLOW):
                clock_cap *= 2
            if over & (_abi.FLAG_POOL_OVERFLOW | _abi.FLAG_FIFO_OVERFLOW):
                pool_overflows += 1
                if pool_overflows == 1:      # an estimate that was a little short: the flagged pool, fourfold
                    if over & _abi.FLAG_POOL_OVERFLOW:
                        cap = min(_abi.MAX_REQUEST_CAPACITY, cap * 4)
                    if over & _abi.FLAG_FIFO_OVERFLOW:
                        fifo = _fifo_pow2(fifo * 4)
                else:
                    # twice: a server the estimate did not see as saturated -- its backlog grows with the horizon.  The live
                    # requests of a scenario never exceed its arrivals (clock_cap bounds them), the waiters of a queue never
                    # the live requests: go to those bounds at once instead of re-running the sweep for every factor of four
                    # (pool and queue overflow in turns: found by scripts/gpu_fuzz_sweeps.py, six runs were not enough).
                    # Round 6: a wait queue may hold up to 2^20 requests (the reference's simpy queues have no bound at all,
                    # server.py:146-149, 210-227) -- 32 bytes of HBM per slot, server and queue pair; the engine runs a sweep
                    # whose state does not fit the device in pieces.
                    cap = min(_abi.MAX_REQUEST_CAPACITY, max(cap * 4, (clock_cap + 7) // 8 * 8))
                    fifo = _fifo_pow2(max(fifo * 4, clock_cap))
                # (a request waits in a queue or for a timed event: the pool need not hold more than 65 535 of the latter)
                cap = min(_abi.MAX_REQUEST_CAPACITY, max(cap, fifo))
            warnings.warn(
                f"engine capacity overflow (flags={over:#x}); retrying with request_capacity={cap}, "
                f"fifo_capacity={fifo}, clock_capacity={clock_cap}", RuntimeWarning, stacklevel=2)
            del counts, clock, samples, res, online_hist, online_rps, summ_out
        res.raise_on_overflow()
        if self.on_negative_delay == "raise":
            res.raise_on_negative_delay()
        return res[0] if self._single else res

    @classmethod
    def from_yaml(cls, *, env: Any = None, yaml_path: str | Path, **kwargs: Any) -> "SimulationRunner":
        """``SimulationRunner.from_yaml`` (simulation_runner.py:381-398)."""
        return cls(env=env, simulation_input=load_yaml(yaml_path), **kwargs)
