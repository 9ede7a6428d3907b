"""Time the UNMODIFIED Python reference (/root/reference, SimulationRunner.run + get_latency_stats) on BASELINE config 2.

Build-container only (the GPU box has no /root/reference): the result is committed as profiles/rNN/python_reference.json and
carried by bench.py into `cpu_baseline.python_reference` with `measured_in_this_run: false`.

    python scripts/measure_python_reference.py profiles/r03/python_reference.json [--replicas 8] [--procs 8]

One replica alone on one core, then `procs` replicas at once (one per process).  request-events per replica: the mean of
the C oracle over the same number of seeds of the same payload (the reference does not count them; its own RNG stream
differs from the engine's, the workload is the same: 78 966 +- 1 300 arrivals per replica).
"""
from __future__ import annotations

import argparse
import json
import multiprocessing as mp
import os
import platform
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def _one(seed: int) -> tuple[float, int]:
    from asyncflow_amd.workloads import lb_two_servers
    from oracle.reference_runner import run_reference_numpy

    t0 = time.perf_counter()
    an = run_reference_numpy(lb_two_servers(), seed)
    st = an.get_latency_stats()
    return time.perf_counter() - t0, int(list(st.values())[0])


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--replicas", type=int, default=8)
    ap.add_argument("--procs", type=int, default=min(8, os.cpu_count() or 1))
    args = ap.parse_args()
    from asyncflow_amd.plan import lower
    from asyncflow_amd.workloads import lb_two_servers
    from oracle import oracle_lib as ol
    from oracle import ref_env

    ref_env.install()
    plan = lower(lb_two_servers())
    ev = [ol.simulate(plan, 0x5EED0000 + i, want_clock=False, want_samples=False).events for i in range(args.replicas)]
    ev_mean = sum(ev) / len(ev)
    alone_s, completed = _one(0)
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(args.procs) as pool:
        parts = pool.map(_one, range(1, 1 + args.replicas), chunksize=1)
    wall = time.perf_counter() - t0
    cpu = ""
    try:
        cpu = next(line.split(":", 1)[1].strip() for line in open("/proc/cpuinfo") if line.startswith("model name"))
    except (OSError, StopIteration):
        cpu = platform.processor()
    out = {
        "what": "unmodified reference: SimulationRunner(env, payload).run() + ResultsAnalyzer.get_latency_stats(), "
                "runner.rng = np.random.default_rng(seed) (numpy PCG64)",
        "workload": "two_servers_lb.yml, T = 600 s, one replica per process",
        "where": "build container",
        "cpu_model": cpu,
        "python": platform.python_version(),
        "simpy_flavour": ref_env.simpy_flavour(),
        "request_events_per_replica": ev_mean,
        "completed_first_replica": completed,
        "one_replica_alone_s": alone_s,
        "value_single_core": ev_mean / alone_s,
        "replicas": args.replicas,
        "procs": args.procs,
        "pool_wall_s": wall,
        "mean_replica_s_loaded": sum(p[0] for p in parts) / len(parts),
        "value_all_procs": ev_mean * args.replicas / wall,
        "replicas_per_s_all_procs": args.replicas / wall,
        "extrapolated_10k_sweep_hours": 10_000 / (args.replicas / wall) / 3600.0,
        "unit": "request-events/s",
    }
    Path(args.out).write_text(json.dumps(out, indent=1))
    print(json.dumps(out, indent=1))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
