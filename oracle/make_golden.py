"""Generate tests/golden/*.npz by running the UNMODIFIED reference here.

ORACLE / TEST INFRASTRUCTURE ONLY.  Build-container only (needs /root/reference):

    python oracle/make_golden.py            # (re)write every fixture
    python oracle/make_golden.py --check    # regenerate in memory, compare with committed files

Each fixture holds, for one (payload, seed) of oracle/scenarios.py::GOLDEN:

* ``payload_json``   the exact input dict (so the fixture is self-contained);
* ``seed``           Philox key of the scenario;
* ``generated / completed / dropped / ticks / heap_events``;
* ``clock``          float64 [completed, 2] = ClientRuntime.rqs_clock (start, finish);
* ``samples``        uint32 [n_series, ticks] sampled series (ram rows: float32 bits);
* ``ram_f64``        (fixtures whose endpoints need FRACTIONAL megabytes only) float64 [n_servers, ticks]: ram_in_use as the
                     reference's collector holds it -- the engine emits the series as float32, i.e. this value rounded once;
* ``latency_stats``  the 8 numbers of ResultsAnalyzer.get_latency_stats();
* ``rps``            ResultsAnalyzer.get_throughput_series()[1];
* ``glibc_log_*``    the same run WITHOUT the math.log substitution (see
                     oracle/reference_runner.py::_DeterministicMath): counts and the
                     max |delta| of (start, finish) -- documents the <= 1 ulp/log effect.

The reference ran on: the SimPy stand-in (oracle/simpy_standin) unless a real
SimPy is importable -- recorded in ``simpy_flavour``.
"""

from __future__ import annotations

import argparse
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from oracle import ref_env  # noqa: E402
from oracle.scenarios import GOLDEN  # noqa: E402

GOLDEN_DIR = ROOT / "tests" / "golden"
STAT_KEYS = ("total_requests", "mean", "median", "std_dev", "p95", "p99", "min", "max")


def build_fixture(name: str) -> dict[str, np.ndarray]:
    from oracle.reference_runner import run_reference

    builder, seed = GOLDEN[name]
    payload = builder()
    res = run_reference(payload, seed, patch_log=True)
    raw = run_reference(payload, seed, patch_log=False)
    n = min(len(res.clock), len(raw.clock))
    delta = float(np.max(np.abs(res.clock[:n] - raw.clock[:n]))) if n else 0.0
    extra = {}
    if np.any(res.ram_f64 != np.floor(res.ram_f64)):
        extra["ram_f64"] = res.ram_f64
    return {
        **extra,
        "payload_json": np.array(json.dumps(payload, sort_keys=True)),
        "seed": np.uint64(seed),
        "generated": np.int64(res.generated),
        "completed": np.int64(res.completed),
        "dropped": np.int64(res.dropped),
        "ticks": np.int64(res.ticks),
        "heap_events": np.int64(res.heap_events),
        "clock": res.clock,
        "samples": res.samples,
        "latency_stats": np.asarray([res.latency_stats.get(k, np.nan) for k in STAT_KEYS], dtype=np.float64),
        "rps": np.asarray(res.throughput[1], dtype=np.float64),
        "edge_ids": np.array(json.dumps(res.edge_ids)),
        "server_ids": np.array(json.dumps(res.server_ids)),
        "simpy_flavour": np.array(res.simpy_flavour),
        "glibc_log_counts": np.asarray([raw.generated, raw.completed, raw.dropped, raw.ticks], dtype=np.int64),
        "glibc_log_max_abs_delta": np.float64(delta),
    }


POOLED_PATH = GOLDEN_DIR / "pooled_lb2_numpy.npz"
POOLED_BINS, POOLED_MAX = 25_600, 0.256      # 10-us latency bins, last bin takes the overflow


def _pooled_worker(seed: int) -> dict[str, np.ndarray]:
    """One replica of two_servers_lb.yml on the STOCK reference: numpy PCG64 seeded through the
    `runner.rng` seam of /root/reference/tests/integration/single_server/test_int_single_server.py:36."""
    from asyncflow_amd.workloads import lb_two_servers
    from oracle.reference_runner import run_reference_numpy

    an = run_reference_numpy(lb_two_servers(), seed)
    st = {str(getattr(k, "value", k)): float(v) for k, v in an.get_latency_stats().items()}
    lat = np.asarray(an.latencies, dtype=np.float64)
    hist = np.bincount(np.minimum((lat * (POOLED_BINS / POOLED_MAX)).astype(np.int64), POOLED_BINS - 1), minlength=POOLED_BINS)
    rps = np.asarray(an.get_throughput_series()[1], dtype=np.float64)
    sm = an.get_sampled_metrics()
    series = {f"{m}:{ent}": float(np.mean(v)) for m, d in sm.items() for ent, v in d.items()}
    return {
        "stats": np.asarray([st[k] for k in STAT_KEYS], dtype=np.float64),
        "hist": hist.astype(np.int64),
        "thin": lat[500::1000].copy(),     # every 1000th completion: ~7.5 s apart, practically independent
        "rps_mean": np.float64(rps.mean()),
        "series_keys": np.array(json.dumps(sorted(series))),
        "series_mean": np.asarray([series[k] for k in sorted(series)], dtype=np.float64),
    }


def build_pooled(n_seeds: int, procs: int) -> None:
    """SURVEY 8d criterion (2): pooled statistics of >= 256 numpy-seeded reference replicas of LB-2."""
    import multiprocessing as mp

    with mp.get_context("fork").Pool(procs) as pool:
        parts = pool.map(_pooled_worker, range(n_seeds), chunksize=1)
    np.savez_compressed(
        POOLED_PATH,
        n_seeds=np.int64(n_seeds),
        seeds=np.arange(n_seeds, dtype=np.int64),
        stats=np.stack([p["stats"] for p in parts]),
        hist=np.sum([p["hist"] for p in parts], axis=0),
        hist_bins=np.int64(POOLED_BINS), hist_max=np.float64(POOLED_MAX),
        thin=np.concatenate([p["thin"] for p in parts]),
        rps_mean=np.asarray([p["rps_mean"] for p in parts]),
        series_keys=parts[0]["series_keys"],
        series_mean=np.stack([p["series_mean"] for p in parts]),
        simpy_flavour=np.array(ref_env.simpy_flavour()),
        generator=np.array("numpy PCG64 via runner.rng = np.random.default_rng(seed); unmodified reference actors"),
    )
    print(f"pooled: {n_seeds} replicas -> {POOLED_PATH.name} ({POOLED_PATH.stat().st_size} B)")


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--pooled", type=int, default=0, metavar="N",
                    help="write tests/golden/pooled_lb2_numpy.npz from N numpy-seeded reference replicas of LB-2 "
                         "(T = 600 s; ~12 s of one core each)")
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("names", nargs="*")
    args = ap.parse_args()
    ref_env.install()
    GOLDEN_DIR.mkdir(parents=True, exist_ok=True)
    if args.pooled:
        build_pooled(args.pooled, args.procs)
        return 0
    rc = 0
    for name in args.names or list(GOLDEN):
        fx = build_fixture(name)
        path = GOLDEN_DIR / f"{name}.npz"
        if args.check:
            old = np.load(path, allow_pickle=False)
            same = all(k in old.files and np.array_equal(old[k], fx[k]) for k in fx)
            print(f"{name}: {'identical' if same else 'DIFFERENT'}")
            rc |= 0 if same else 1
        else:
            np.savez_compressed(path, **fx)
            print(
                f"{name}: generated={int(fx['generated'])} completed={int(fx['completed'])} "
                f"dropped={int(fx['dropped'])} ticks={int(fx['ticks'])} heap_events={int(fx['heap_events'])} "
                f"glibc-log delta={float(fx['glibc_log_max_abs_delta']):.3e} -> {path.name} ({path.stat().st_size} B)"
            )
    return rc


if __name__ == "__main__":
    raise SystemExit(main())
