"""Run the UNMODIFIED reference (imported from /root/reference) on the spec streams.

ORACLE / TEST INFRASTRUCTURE ONLY.  Build-container only: /root/reference does
not exist on the GPU box; what travels are the fixtures this produces
(oracle/make_golden.py -> tests/golden/).

``run_reference`` executes exactly the statements of ``SimulationRunner.run``
(/root/reference/src/asyncflow/runtime/simulation_runner.py:349-376) with ONE
addition between the build and the start phase: each actor's ``rng`` attribute
is replaced by a per-actor stream adapter (the same private-method recipe as the
reference's tests/integration/minimal/test_minimal.py:69-84).
"""

from __future__ import annotations

import types
from dataclasses import dataclass, field
from typing import Any

import numpy as np

from . import oracle_lib as ol
from . import ref_env
from .rng_adapter import EdgeRNG, GeneratorRNG, ServerRNG


@dataclass
class ReferenceResult:
    generated: int
    completed: int
    dropped: int
    ticks: int
    heap_events: int                  # SimPy heap pushes (incl. zero-time plumbing)
    clock: np.ndarray                 # float64 [completed, 2]
    samples: np.ndarray               # uint32 [n_series, ticks] (ram rows: float32 bits)
    latency_stats: dict[str, float]
    throughput: tuple[list[float], list[float]]
    edge_ids: list[str] = field(default_factory=list)
    server_ids: list[str] = field(default_factory=list)
    simpy_flavour: str = "standin"
    ram_f64: np.ndarray = field(default_factory=lambda: np.zeros((0, 0)))   # float64 [n_servers, ticks]: ram_in_use as the reference holds it


class _DeterministicMath(types.SimpleNamespace):
    """Stand-in for the `math` module inside the sampler modules.

    samplers/poisson_poisson.py:70 and gaussian_poisson.py:82 call ``math.log``
    (glibc, < 1 ulp, not correctly rounded).  For bit-exact comparison with the
    GPU the golden runs route that ONE call to the spec's log (also < 1 ulp);
    ``patch_log=False`` keeps glibc's and is compared with a 1e-9 tolerance
    (tests/test_reference_parity.py).
    """


def _to_words(values: list[Any], *, as_float: bool) -> np.ndarray:
    if as_float:
        return np.asarray(values, dtype=np.float64).astype(np.float32).view(np.uint32)
    return np.asarray(values, dtype=np.int64).astype(np.int32).view(np.uint32)


def run_reference(payload: dict, seed: int, *, patch_log: bool = True) -> ReferenceResult:
    ref_env.install()
    import math as real_math

    import simpy
    from asyncflow.config.constants import SampledMetricName
    from asyncflow.metrics.analyzer import ResultsAnalyzer
    from asyncflow.runtime.simulation_runner import SimulationRunner
    from asyncflow.samplers import gaussian_poisson, poisson_poisson
    from asyncflow.schemas.payload import SimulationPayload

    model = SimulationPayload.model_validate(payload)
    env = simpy.Environment()
    heap_events = 0
    _schedule = env.schedule

    def counting_schedule(*a: Any, **k: Any) -> None:
        nonlocal heap_events
        heap_events += 1
        _schedule(*a, **k)

    env.schedule = counting_schedule  # type: ignore[method-assign]
    runner = SimulationRunner(env=env, simulation_input=model)

    saved = (poisson_poisson.math, gaussian_poisson.math)
    if patch_log:
        shim = _DeterministicMath(log=ol.lib().orc_x_log)
        poisson_poisson.math = shim  # type: ignore[assignment]
        gaussian_poisson.math = shim  # type: ignore[assignment]
    try:
        # ---- simulation_runner.py:351-361 (build, wire, attach events)
        runner._build_rqs_generator()  # noqa: SLF001
        runner._build_client()  # noqa: SLF001
        runner._build_servers()  # noqa: SLF001
        runner._build_load_balancer()  # noqa: SLF001
        runner._build_edges()  # noqa: SLF001
        runner._build_events()  # noqa: SLF001

        # ---- the ONE addition: per-actor spec streams
        for gen in runner._rqs_runtime.values():  # noqa: SLF001
            gen.rng = GeneratorRNG(seed)
        edge_rngs = []
        for i, e in enumerate(runner.edges):
            rt = runner._edges_runtime[(e.source, e.target)]  # noqa: SLF001
            rt.rng = EdgeRNG(seed, i)
            edge_rngs.append(rt.rng)
        for i, s in enumerate(runner.servers):
            runner._servers_runtime[s.id].rng = ServerRNG(seed, i)  # noqa: SLF001

        # ---- simulation_runner.py:364-369 (start, run)
        runner._start_events()  # noqa: SLF001
        runner._start_all_processes()  # noqa: SLF001
        runner._start_metric_collector()  # noqa: SLF001
        env.run(until=runner.simulation_settings.total_simulation_time)
    finally:
        poisson_poisson.math, gaussian_poisson.math = saved
    assert poisson_poisson.math is real_math

    client = next(iter(runner._client_runtime.values()))  # noqa: SLF001
    servers = list(runner._servers_runtime.values())  # noqa: SLF001
    edges = list(runner._edges_runtime.values())  # noqa: SLF001
    analyzer = ResultsAnalyzer(client=client, servers=servers, edges=edges, settings=runner.simulation_settings)

    clock = np.asarray([(c.start, c.finish) for c in client.rqs_clock], dtype=np.float64).reshape(-1, 2)
    rows = []
    conn_key = SampledMetricName.EDGE_CONCURRENT_CONNECTION
    for e in edges:
        rows.append(_to_words(e.enabled_metrics.get(conn_key, []), as_float=False))
    for s in servers:
        m = s.enabled_metrics
        rows.append(_to_words(m.get(SampledMetricName.READY_QUEUE_LEN, []), as_float=False))
        rows.append(_to_words(m.get(SampledMetricName.EVENT_LOOP_IO_SLEEP, []), as_float=False))
        rows.append(_to_words(m.get(SampledMetricName.RAM_IN_USE, []), as_float=True))
    ticks = max((len(r) for r in rows), default=0)
    samples = np.zeros((len(rows), ticks), dtype=np.uint32)
    for i, r in enumerate(rows):
        samples[i, : len(r)] = r
    # ram_in_use exactly as the reference's collector appended it (int | float, server.py:65): the engine emits it as f32
    ram_f64 = np.zeros((len(servers), ticks), dtype=np.float64)
    for i, srv in enumerate(servers):
        vals = srv.enabled_metrics.get(SampledMetricName.RAM_IN_USE, [])
        ram_f64[i, : len(vals)] = np.asarray(vals, dtype=np.float64)

    stats = {str(getattr(k, "value", k)): float(v) for k, v in analyzer.get_latency_stats().items()}
    gen = next(iter(runner._rqs_runtime.values()))  # noqa: SLF001
    return ReferenceResult(
        generated=int(gen.id_counter),
        completed=len(client.rqs_clock),
        dropped=sum(r.sends - r.latencies for r in edge_rngs),
        ticks=ticks,
        heap_events=heap_events,
        clock=clock,
        samples=samples,
        ram_f64=ram_f64,
        latency_stats=stats,
        throughput=analyzer.get_throughput_series(),
        edge_ids=[e.edge_config.id for e in edges],
        server_ids=[s.server_config.id for s in servers],
        simpy_flavour=ref_env.simpy_flavour(),
    )


def run_reference_numpy(payload: dict, seed: int) -> Any:
    """Stock reference: numpy PCG64 seeded through ``runner.rng`` (statistical parity)."""
    ref_env.install()
    import simpy
    from asyncflow.runtime.simulation_runner import SimulationRunner
    from asyncflow.schemas.payload import SimulationPayload

    model = SimulationPayload.model_validate(payload)
    runner = SimulationRunner(env=simpy.Environment(), simulation_input=model)
    runner.rng = np.random.default_rng(seed)
    return runner.run()
