"""Benchmark of the hot path: batched `env.run(until=T)` on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C] [--scenarios S]

Workload (BASELINE.json configs[1], the one the metric is quoted on, default): the 2-server +
load-balancer topology of examples/yaml_input/data/two_servers_lb.yml (400 users x 20 rpm,
T = 600 s, 0.05 s sampling), 10 000 seed replicas per GPU, scenario i uses Philox key
0x5EED0000 + i, full-fidelity outputs (every (start, finish) pair and all 12 sampled series
written to HBM).  The other BASELINE configs are available with --config:
  1  single_server.yml, T = 300 s, ONE replica (the reference's own CPU-runnable case)
  3  users x RTT 100 x 100 grid (10 000 scenarios per GPU)
  4  the grid x 10 seeds = 100 000 scenarios with event_inj_lb.yml's spikes / outages, SHARDED over
     the ranks by expected load (strong scaling: the stated total is split, not replicated)
  5  8-server fan-out with log-normal edges, 50 000 replicas sharded over the ranks
  6  (not a BASELINE config) LB-2 with two endpoints per server and core re-entry, 10 000 replicas per GPU: the
     workload of the general server station (server.py:79-313 without the tandem restriction)

A "step" is ONE pass of the hot path over the rank's batch: af_engine_run() (seed upload, the HIP
kernels, stream sync) + af_engine_summarize() (the batched analyzer), inputs/outputs resident in
HBM; batches that do not fit the HBM budget run as slices that reuse the output buffers.  K steps
are timed between barrier + torch.cuda.synchronize() on both sides, MAX over ranks; rank 0 prints
ONE JSON line.  value = request-events simulated by all ranks / that time (request-event = one
timed state transition of a request: arrival, edge delivery, CPU-step end, I/O-step end;
SURVEY.md section 8d).

N > 1: one process per GPU.  Under torchrun (RANK / WORLD_SIZE in the environment) this process IS
a rank; called plainly with --gpus N it re-executes itself under `python -m torch.distributed.run
--nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`.  Scenarios are sharded by rank with NO
data-path collective; ONE all-gather of the per-scenario summaries over xGMI after the timed region
(reported as gather_ms).  `--selftest-cpu` drives the same launcher / sharding / gather path on CPU
(gloo) with synthetic per-scenario summaries and no engine (tests/test_distributed_cpu.py).
"""

from __future__ import annotations

import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6290 GB/s measured-achievable
HBM_ACHIEVABLE_GBS = 6290.0
CONFIG_TOTALS = {4: 100_000, 5: 50_000}     # BASELINE.json: totals stated for the 8-GPU configs


# --------------------------------------------------------------------------- #
# workloads (asyncflow_amd/workloads.py holds the BASELINE payloads)            #
# --------------------------------------------------------------------------- #
def build_workload(cfg: int, rank: int, world: int, scenarios: int, horizon: int | None) -> dict:
    """The rank's share of BASELINE config `cfg`: payload, seeds, per-scenario parameter columns."""
    from asyncflow_amd import workloads as w
    from asyncflow_amd.distributed import interleave_by_load, shard_bounds

    base = w.BASELINE_SEED_BASE[cfg]
    cols: dict[str, np.ndarray] = {}
    scaling = "weak"
    if cfg == 1:
        T = horizon or 300
        payload, n = w.single_server(horizon=T), scenarios or 1
        label = f"single_server.yml (100 users x 20 rpm, T={T} s), {n} replica(s) per GPU"
        seeds = base + rank * n + np.arange(n, dtype=np.uint64)
    elif cfg == 2:
        T = horizon or 600
        payload, n = w.lb_two_servers(horizon=T), scenarios or 10_000
        label = f"two_servers_lb.yml (2 servers + LB, 400 users x 20 rpm, T={T} s, dt=0.05 s), {n} seed replicas per GPU"
        seeds = base + rank * n + np.arange(n, dtype=np.uint64)
    elif cfg == 3:
        T = horizon or 600
        payload = w.lb_two_servers(horizon=T)
        side = int(round(math.sqrt(scenarios or 10_000)))
        n = side * side
        a, b = w.grid_users_rtt(side)
        order = np.argsort(-a, kind="stable")            # heaviest first: similar scenarios run together
        cols = {"rqs_input.avg_active_users.mean": a[order], "topology_graph.edges[*].latency.mean": b[order]}
        seeds = (base + rank * n + np.arange(n, dtype=np.uint64))[order]
        label = f"LB-2 grid avg_active_users (10..1000) x per-hop latency (0.5..50 ms), {side}x{side} points per GPU, T={T} s"
    elif cfg == 4:
        T = horizon or 600
        payload = w.lb_with_events(users=400, horizon=T, scale=T / 600.0)
        total = scenarios or CONFIG_TOTALS[4]
        reps = max(1, total // 10_000)
        side = int(round(math.sqrt(total / reps)))
        a, b = w.grid_users_rtt(side)
        a, b = np.repeat(a, reps), np.repeat(b, reps)
        total = a.size
        all_seeds = base + np.arange(total, dtype=np.uint64)
        mine = interleave_by_load(a, world)[rank]         # deal by expected event count
        mine = mine[np.argsort(-a[mine], kind="stable")]
        cols = {"rqs_input.avg_active_users.mean": a[mine], "topology_graph.edges[*].latency.mean": b[mine]}
        seeds, n, scaling = all_seeds[mine], int(mine.size), "strong"
        label = (f"LB-2 grid {side}x{side} x {reps} seeds = {total} scenarios with event_inj_lb.yml spikes/outages, "
                 f"T={T} s, sharded by expected load over {world} GPU(s)")
    elif cfg == 5:
        T = horizon or 600
        payload = w.fanout8(horizon=T)
        total = scenarios or CONFIG_TOTALS[5]
        lo, hi = shard_bounds(total, rank, world)
        seeds, n, scaling = base + np.arange(lo, hi, dtype=np.uint64), hi - lo, "strong"
        label = f"8-server fan-out, log-normal edges (120 users x 20 rpm, T={T} s), {total} replicas sharded over {world} GPU(s)"
    elif cfg == 6:
        T = horizon or 600
        payload, n = w.lb_two_servers_two_endpoints(horizon=T), scenarios or 10_000
        label = (f"NOT a BASELINE config: LB-2 with two endpoints per server and core re-entry (general servers), 400 users x 20 rpm, "
                 f"T={T} s, {n} seed replicas per GPU")
        seeds = base + rank * n + np.arange(n, dtype=np.uint64)
    else:
        raise ValueError(cfg)
    return {"payload": payload, "seeds": np.ascontiguousarray(seeds, dtype=np.uint64), "columns": cols, "n": int(n),
            "label": label, "scaling": scaling, "horizon": T}


def algorithmic_bytes(counts: np.ndarray, n_series: int, plan_bytes: int) -> float:
    """B = 80 N_events + 16 N_completed + 4 n_series N_ticks + plan_bytes, per scenario."""
    ev = counts[:, 3].astype(np.float64).sum()
    comp = counts[:, 1].astype(np.float64).sum()
    ticks = counts[:, 4].astype(np.float64).sum()
    return 80.0 * ev + 16.0 * comp + 4.0 * n_series * ticks + float(plan_bytes) * counts.shape[0]


# --------------------------------------------------------------------------- #
# CPU baseline legs (oracle / hostcheck / reference are CHECKERS, timed beside) #
# --------------------------------------------------------------------------- #
def cgroup_cpu_quota() -> float | None:
    """CPUs the cgroup lets this process use (cpu.max / cfs quota), or None when unlimited."""
    try:
        text = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if text and text[0] != "max":
            return float(text[0]) / float(text[1])
    except (OSError, ValueError, IndexError):
        pass
    try:
        q = float(Path("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read_text())
        p = float(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
        if q > 0 and p > 0:
            return q / p
    except (OSError, ValueError):
        pass
    return None


def _cpu_worker(args: tuple[dict, list[int]]) -> tuple[int, int, float]:
    payload, seeds = args
    from asyncflow_amd.plan import lower
    from oracle import oracle_lib as ol

    plan = lower(payload)
    ol.simulate(plan, 1, want_clock=True, want_samples=True)  # warm
    t0 = time.perf_counter()
    ev = heap = 0
    for s in seeds:
        r = ol.simulate(plan, s)
        ev += r.events
        heap += r.heap_events
    return ev, heap, time.perf_counter() - t0


def _time_serial(fn, budget_s: float) -> tuple[float, int, float]:
    """Call fn(i) -> events until budget_s of wall is used; (events, calls, seconds)."""
    ev, k, t0 = 0, 0, time.perf_counter()
    while True:
        ev += fn(k)
        k += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s:
            return float(ev), k, dt


def committed_python_reference() -> dict | None:
    """The newest profiles/rNN/python_reference.json: the UNMODIFIED Python reference timed in the build container."""
    found = sorted((ROOT / "profiles").glob("r*/python_reference.json"))
    if not found:
        return None
    pj = json.loads(found[-1].read_text())
    return {"value": pj["value_single_core"], "unit": "request-events/s", "cores": 1, "kind": "reference",
            "measured_in_this_run": False, "where": pj["where"], "cpu_model": pj["cpu_model"], "simpy_flavour": pj["simpy_flavour"],
            "all_cores": {"value": pj["value_all_procs"], "cores": pj["procs"], "replicas_per_s": pj["replicas_per_s_all_procs"],
                          "extrapolated_10k_sweep_hours": pj["extrapolated_10k_sweep_hours"]},
            "sample": f"{pj['workload']}; {pj['what']}; one replica alone {pj['one_replica_alone_s']:.2f} s",
            "source": str(found[-1].relative_to(ROOT)) + " (scripts/measure_python_reference.py)"}


def cpu_baseline(payload: dict, seed_base: int, label: str, budget_s: float = 8.0) -> dict:
    """CPU legs timed in THIS run on the host cores of this box, on a bounded sample of the same workload:
    (a) the SimPy-faithful C restatement (oracle/des_oracle.c) on ONE unloaded core,
    (b) the same on every usable core (affinity capped by the cgroup quota) -- the reported `value`,
    (c) `port_lean`: the engine's own next-event core compiled for the host (tests/hostcheck), one core,
    (d) the unmodified Python reference (numpy-seeded, runner.rng seam) when /root/reference exists here.
    """
    import multiprocessing as mp

    from asyncflow_amd import _abi
    from asyncflow_amd.plan import lower
    from oracle import oracle_lib as ol
    from oracle import ref_env

    ol.build()
    plan = lower(payload)
    ol.simulate(plan, seed_base)                                    # warm (page in, caches)
    heap_total = [0]

    def one(i: int) -> int:
        r = ol.simulate(plan, seed_base + i)
        heap_total[0] += r.heap_events
        return r.events

    ev1, runs1, s1 = _time_serial(one, min(3.0, budget_s / 2))
    single = ev1 / s1
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = cgroup_cpu_quota()
    procs = max(1, min(affinity, int(math.ceil(quota)) if quota else affinity))
    per_run = s1 / runs1
    per_core = max(2, min(32, int(budget_s / (3.0 * per_run))))     # (a run is ~2-5x slower with every hardware thread busy)
    jobs = [(payload, [seed_base + 1000 + c * per_core + k for k in range(per_core)]) for c in range(procs)]
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(procs) as pool:
        parts = pool.map(_cpu_worker, jobs, chunksize=1)
    wall = time.perf_counter() - t0
    ev = float(sum(p[0] for p in parts))
    heap = float(sum(p[1] for p in parts))
    busy = max(p[2] for p in parts)
    out = {
        "value": ev / busy,
        "unit": "request-events/s",
        "cores": procs,
        "kind": "port",
        "sample": f"{procs * per_core} scenarios of the same workload ({label}), {per_core} per process on {procs} processes, "
                  f"C restatement of the reference actors + SimPy heap (oracle/des_oracle.c), "
                  f"{heap / max(ev, 1.0):.1f} SimPy heap events per request-event, pool wall {wall:.1f} s",
        "scenarios_per_s": procs * per_core / busy,
        "single_core_unloaded": single,
        "per_core_loaded": ev / busy / procs,
        "cores_affinity": affinity,
        "cores_cgroup_quota": quota,
    }
    ratio = out["per_core_loaded"] / single
    if ratio < 0.5:
        out["note"] = (f"per-core rate with {procs} busy processes is {ratio:.2f}x the unloaded single-core rate: the "
                       f"'cores' of this box are SMT threads / oversubscribed vCPUs (sched_getaffinity={affinity}, "
                       f"cgroup quota={quota}); single_core_unloaded is the number comparable across machines")
    # (c) the engine's own lean core on the host
    try:
        from tests.hostcheck import build as hc

        hc.build()
        hc.simulate(plan, seed_base)

        def lean(i: int) -> int:
            counts, _, _ = hc.simulate(plan, seed_base + i)
            return int(counts[_abi.CNT_EVENTS])

        evl, _, sl = _time_serial(lean, min(2.5, budget_s / 3))
        out["port_lean"] = {"value": evl / sl, "unit": "request-events/s", "cores": 1, "kind": "port-lean",
                            "what": "asyncflow_amd/csrc/af_core.hpp (the sequential next-event core) compiled by g++ for one lane"}
    except Exception as exc:  # noqa: BLE001 - a missing g++ must not take the bench down
        out["port_lean"] = {"value": None, "why": f"{type(exc).__name__}: {exc}"}
    # (d) the Python reference itself: timed HERE only where it exists (never on the GPU box); otherwise the committed
    # build-container measurement of scripts/measure_python_reference.py is carried, labelled as such
    out["python_reference"] = committed_python_reference()
    if ref_env.reference_available():
        try:
            from oracle.reference_runner import run_reference_numpy

            short = json.loads(json.dumps(payload))
            T_full = float(short["sim_settings"]["total_simulation_time"])
            T_short = max(5.0, min(T_full, 60.0))
            short["sim_settings"]["total_simulation_time"] = int(T_short)
            t0 = time.perf_counter()
            an = run_reference_numpy(short, 0)
            an.get_latency_stats()
            dt = time.perf_counter() - t0
            ev_short = ev1 / runs1 * (T_short / T_full)        # request-events of the same scenario length
            out["python_reference"] = {"value": ev_short / dt, "unit": "request-events/s", "cores": 1, "kind": "reference",
                                       "measured_in_this_run": True, "where": "this box",
                                       "sample": f"1 replica, T={int(T_short)} s, unmodified reference actors on "
                                                 f"{ref_env.simpy_flavour()} SimPy, numpy PCG64 via runner.rng, wall {dt:.2f} s",
                                       "committed": committed_python_reference()}
        except Exception as exc:  # noqa: BLE001
            out["python_reference"] = {"value": None, "why": f"{type(exc).__name__}: {exc}"}
    return out


def kernel_sources_sha1() -> str:
    import hashlib

    h = hashlib.sha1()
    for name in ("engine.hip", "af_flow.hpp", "af_flow_host.hpp", "af_core.hpp", "af_math.hpp", "af_plan_pack.hpp", "af_summary.hpp", "af_pregen.hpp"):
        h.update((ROOT / "asyncflow_amd" / "csrc" / name).read_bytes())
    return h.hexdigest()


BINDING_LABEL = {2: "c2", 3: "c3", 4: "c4", 5: "c5", 6: "gensrv"}


def find_binding(config: int) -> Path | None:
    """Newest committed profiles/rNN/binding_<label>.json of `--config` (config 2 also answers to rounds 3-4's binding.json)."""
    label = BINDING_LABEL.get(config)
    if label is None:
        return None
    found = sorted((ROOT / "profiles").glob(f"r*/binding_{label}.json"))
    if not found and config == 2:
        found = sorted((ROOT / "profiles").glob("r*/binding.json"))
    return found[-1] if found else None


def attach_binding(roof: dict, dom_ms: float, launches: int, args, n: int, wl: dict) -> None:
    """What limits the dominant kernel and what HBM really moved, from the committed PMC passes of THIS command (rocprofv3
    cannot run inside the timed bench): profiles/rNN/binding_<config>.json (scripts/profile_round5.sh + make_binding_json.py).
    `stale` says whether the kernel sources changed since the profile was taken; times are this run's.  Sets
    roof["traffic"] (HBM bytes per step = per-launch counter bytes x launches of the step) and roof["binding"]."""
    default_n = {2: 10_000, 3: 10_000, 4: CONFIG_TOTALS[4], 5: CONFIG_TOTALS[5], 6: 10_000}.get(args.config)
    if not (default_n == (n if args.config in (2, 3, 6) else args.scenarios or default_n) and not args.no_series and wl["horizon"] == 600
            and not args.online_summary and args.gpus == 1):
        roof["binding"] = {"why": "PMC passes are committed for the default single-GPU command of each config only"}
        return
    path = find_binding(args.config)
    if path is None:
        roof["binding"] = {"why": f"no profiles/rNN/binding_{BINDING_LABEL.get(args.config)}.json"}
        return
    bj = json.loads(path.read_text())
    stale = bj.get("sources_sha1") != kernel_sources_sha1()
    same_kernel = bj["kernel"].split("_jit")[0].split("_kernel")[0] in roof["kernel"]
    roof["binding"] = {
        "resource": bj.get("binding", "valu_issue"),
        "frac": bj.get("valu_busy_frac_calibrated", bj.get("valu_issue_frac")),
        "frac_is": bj.get("valu_busy_frac_is", "SQ_ACTIVE_INST_VALU x 4 / (1 024 SIMDs x kernel cycles)"),
        "source": f"{path.relative_to(ROOT)} (rocprofv3 --pmc passes of the same command, one counter set per run)",
        "profiled_kernel": bj["kernel"], "stale": bool(stale or not same_kernel),
        "valu_wave_insts_per_request_event": bj["valu_wave_insts_per_request_event"],
        "all_wave_insts_per_request_event": bj.get("all_wave_insts_per_request_event"),
        "valu_issue_frac_uncalibrated_x4": bj["valu_issue_frac"],
        "valu_calibration": bj.get("valu_calibration"),
        "valu_lane_utilisation": bj["valu_lane_utilisation"],
        "wait_any_frac_of_wave_cycles": bj.get("wait_any_frac_of_wave_cycles"),
        "lds_bank_conflict_frac": bj.get("lds_bank_conflict_frac"),
        "wave_lifetime_ms": bj.get("wave_lifetime_ms"),
        "hbm_read_bytes_per_launch_FETCH_SIZE_x2": bj.get("hbm_read_bytes"), "l2_write_bytes_per_launch_WRITE_SIZE": bj.get("l2_write_bytes"),
        "profiled_kernel_ms": bj.get("kernel_avg_ms_trace"),
    }
    cal = sorted((ROOT / "profiles").glob("r*/write_calibration.json"))
    if cal:
        cj = json.loads(cal[-1].read_text())
        roof["binding"]["write_size_calibration"] = {
            "source": str(cal[-1].relative_to(ROOT)),
            "counter_bytes_per_stored_byte": {k: v.get("counter_bytes_per_stored_byte") for k, v in cj["patterns"].items()},
            "reading": "WRITE_SIZE counts the kernel's store shapes exactly (1.00 B per stored byte): what it shows above the "
                       "output size are the kernel's scratch stores (an L2-level counter) and lines shared by the chunks of "
                       "consecutive rounds, not a counting artefact"}
    if bj.get("hbm_read_bytes") is not None and bj.get("l2_write_bytes") is not None:
        roof["traffic"] = (bj["hbm_read_bytes"] + bj["l2_write_bytes"]) * launches
        roof["traffic_source"] = roof["binding"]["source"]
        roof["traffic_over_algorithmic"] = roof["traffic"] / max(roof["algorithmic_bytes_per_step"], 1.0)
        roof["frac_traffic"] = roof["traffic"] / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS


def end_to_end_sweep(sw: "RankSweep", wl: dict, dev) -> dict:
    """BASELINE.md 4.3's wall-clock of ONE sweep, outside the timed region: lowering the payload, engine creation, the
    step itself (seed / column upload, kernels, analyzer) and the D2H copy of the per-scenario summaries."""
    import torch

    from asyncflow_amd.plan import lower

    t0 = time.perf_counter()
    plan = lower(wl["payload"])
    t_lower = time.perf_counter() - t0
    del plan
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    sw.step()
    torch.cuda.synchronize(dev)
    t_step = time.perf_counter() - t1
    t2 = time.perf_counter()
    host = [sw.s_stats.cpu(), sw.s_rps.cpu(), sw.s_hist.cpu()]
    if sw.s_mean is not None:
        host += [sw.s_mean.cpu(), sw.s_max.cpu()]
    t_d2h = time.perf_counter() - t2
    nbytes = sum(t.numel() * t.element_size() for t in host)
    return {"total_s": t_lower + t_step + t_d2h, "lowering_s": t_lower, "step_s": t_step, "summary_d2h_s": t_d2h,
            "summary_d2h_bytes": nbytes, "jit_build_or_cache_load_s_not_included": sw.jit_build_s,
            "what": "lower(payload) + one step (H2D of seeds / columns, kernels, analyzer) + D2H of the per-scenario summaries; "
                    "the full outputs stay in HBM (their D2H would add 18 GB / PCIe)"}


def residency_tail(sw: "RankSweep", flow_ms: float) -> dict | None:
    """How much the partial last residency round of the one-wave-per-scenario launch costs: the same kernel over the
    largest multiple of the resident wave slots (256 CUs x waves per CU) that fits the batch, per scenario."""
    lds = int(sw.run_stats["flow_lds_bytes"]) or 1
    per_cu = max(1, min(16, (160 * 1024) // ((lds + 511) // 512 * 512)))
    slots = 256 * per_cu
    n = sw.slice if sw.n_slices > 1 else sw.n
    m = (n // slots) * slots
    if m == 0 or m == n:
        return {"waves_per_cu": per_cu, "resident_slots": slots, "rounds": n / slots, "loss_frac": 0.0 if m == n else None}
    seeds, over, kw = sw._slice_args(0, m)   # noqa: SLF001
    st = sw.eng.run(seeds, over, specialise=sw.specialise, **kw)
    full = float(st.flow_kernel_ms) / m
    return {"waves_per_cu": per_cu, "resident_slots": slots, "rounds": n / slots, "full_rounds_scenarios": m,
            "full_rounds_kernel_ms": float(st.flow_kernel_ms), "us_per_scenario_full_rounds": full * 1e3,
            "us_per_scenario_this_batch": flow_ms / n * 1e3, "loss_frac": 1.0 - full * n / flow_ms}


def parity_spot_check(sw: "RankSweep", k: int = 16) -> dict:
    """AFTER the timed region, outside every timing: >= k scenarios of the batch that was just benched, spread over EVERY
    slice of the step -- the first, the last and evenly spaced ones of each slice -- against the CPU oracle
    (oracle/des_oracle.c, the CHECKER, pinned on the reference's fixtures): counts, every (start, finish) pair bit for
    bit, every sample.  A step of several slices reuses its output buffers, so each slice is run again (same seeds,
    same columns, same kernels; untimed) and checked while its outputs are resident.
    Makes the headline line self-certifying: `ok` false = the benched kernel did not compute the reference's results."""
    from asyncflow_amd import _abi
    from asyncflow_amd.plan import lower
    from oracle import oracle_lib as ol

    code_name = {v: name for name, v in _abi.PARAM_CODES.items()}
    per_slice = max(4, -(-k // sw.n_slices))
    bad: list[str] = []
    all_picks: list[int] = []
    t0 = time.perf_counter()
    for s_i, lo in enumerate(range(0, sw.n, sw.slice)):
        hi = min(sw.n, lo + sw.slice)
        m = hi - lo
        if sw.n_slices > 1:            # (the buffers hold the LAST slice of the last step: bring this one back)
            sw.run_slice(lo, hi)
        sw.torch.cuda.synchronize(sw.dev)
        picks = sorted({lo + int(round(j * (m - 1) / max(per_slice - 1, 1))) for j in range(min(per_slice, m))})
        all_picks += picks
        counts = sw.counts.cpu().numpy().view(np.uint32)
        for i in picks:
            plan = lower(sw.plan.payload)
            ol.apply_overrides(plan, {(code_name[c], idx): float(col[i]) for c, idx, col, _ in sw.over})
            want = ol.simulate(plan, int(sw.seeds[i]), clock_capacity=sw.clock_cap, want_clock=sw.clock is not None,
                               want_samples=sw.samples is not None)
            if not np.array_equal(counts[i, :5].astype(np.uint64), want.counts[:5]):
                bad.append(f"scenario {i} (slice {s_i}): counts {counts[i, :5].tolist()} != {want.counts[:5].tolist()}")
                continue
            n_done = int(counts[i, _abi.CNT_COMPLETED])
            if sw.clock is not None:
                got = sw.clock[i - lo, :n_done].cpu().numpy()
                if not np.array_equal(got.view(np.uint64), want.clock.view(np.uint64)):
                    bad.append(f"scenario {i} (slice {s_i}): rqs_clock differs")
            if sw.samples is not None:
                ticks = int(counts[i, _abi.CNT_TICKS])
                got = sw.samples[i - lo, :ticks, : sw.plan.n_series].cpu().numpy().view(np.uint32).T
                if not np.array_equal(got, want.samples):
                    bad.append(f"scenario {i} (slice {s_i}): sampled series differ")
    return {"scenarios": len(all_picks), "indices": all_picks, "slices_covered": sw.n_slices, "ok": not bad, "mismatches": bad[:4],
            "compared": "counts[:5]" + (", rqs_clock (bit patterns)" if sw.clock is not None else "")
                        + (", every sampled series" if sw.samples is not None else ""),
            "checker": "oracle/des_oracle.c (CPU restatement pinned on the reference's fixtures), after the timed region; "
                       "the whole benched batches are compared on the device by tests/test_gpu_full_batches.py",
            "check_s": time.perf_counter() - t0}


# --------------------------------------------------------------------------- #
# the rank's sweep: engine + HBM-resident outputs, run as slices                 #
# --------------------------------------------------------------------------- #
def rank_shape(wl: dict, args) -> dict:
    """Everything about a rank's sweep that does not need a device: the lowered plan, the engine columns, the capacities,
    the slicing and the engine options -- what RankSweep runs and what `prebuild_kernels` asks the specialised kernel for."""
    from asyncflow_amd import _abi
    from asyncflow_amd.plan import estimate_capacities, lower
    from asyncflow_amd.runner import _fifo_pow2, resolve_sweep

    plan = lower(wl["payload"])
    n = wl["n"]
    over = resolve_sweep(plan, wl["columns"], n)
    users = wl["columns"].get("rqs_input.avg_active_users.mean")
    lat = wl["columns"].get("topology_graph.edges[*].latency.mean")
    users_max = float(users.max()) if users is not None else None
    lat_scale = float(lat.max() / plan.edge_mean.min()) if lat is not None else 1.0
    cap, fifo = estimate_capacities(plan, users_max, lat_scale)
    clock_cap = plan.clock_capacity(users_max)
    ticks = max(plan.tick_count, 1)
    # slices: outputs of one slice must fit the HBM budget (clock + samples; the engine's own buffers on top)
    online = bool(args.online_summary)
    if online:
        args.no_series = True
    per_scen = (0 if online else clock_cap * 16) + (0 if args.no_series else ticks * plan.series_pitch * 4) + 8192
    budget = int(args.hbm_budget_gb * (1 << 30))
    slice_n = max(1, min(n, budget // max(per_scen, 1), 65535 if n > 65535 else n))
    n_slices = (n + slice_n - 1) // slice_n
    slice_n = (n + n_slices - 1) // n_slices
    engine_kw = dict(request_capacity=min(cap, _abi.MAX_REQUEST_CAPACITY), fifo_capacity=_fifo_pow2(min(fifo, cap)),
                     lanes_per_wave=args.lanes, force_global_state=args.global_state,
                     expect_shared_instants=args.expect_shared_instants, flow=not args.no_flow,
                     flow_list_entries=args.flow_list_entries,
                     flow_ring_rows=_abi.FLOW_RING_IN_HBM if args.flow_ring_rows < 0 else args.flow_ring_rows)
    return {"plan": plan, "n": n, "seeds": wl["seeds"], "over": over, "clock_cap": clock_cap, "ticks": ticks,
            "T": int(plan.total_time), "online": online, "slice": slice_n, "n_slices": n_slices, "engine_kw": engine_kw}


def prebuild_kernels(configs: tuple[int, ...] = (2, 3, 4, 5, 6), worlds: tuple[int, ...] = (1, 2, 4, 8), verbose: bool = True) -> list[str]:
    """Compile, WITHOUT a GPU, the plan-specialised stage-parallel kernel of every default bench line (`--config C` at
    `--gpus N`) into the JIT cache (asyncflow_amd/csrc/_jit/, which travels to the GPU box with the tree): the headline
    then does not depend on hipcc being present where the bench runs.  A planning-only engine (AF_DEVICE_PLAN_ONLY) answers
    `af_engine_jit_spec` -- a pure function of plan, sweep columns and output shape -- and `jit.code_object` builds what is
    not cached yet.  Called by `__graft_entry__.build()`.  Returns the distinct specs."""
    from asyncflow_amd import jit
    from asyncflow_amd.engine import PLAN_ONLY, Engine

    specs: dict[str, str] = {}
    for cfg in configs:
        for world in worlds:
            if cfg in (2, 3, 6) and world != 1:
                continue            # weak scaling: every rank runs the same shape as the single GPU
            for rank in sorted({0, world - 1}):
                args = make_parser().parse_args(["--config", str(cfg), "--gpus", str(world)])
                args.horizon = None
                wl = build_workload(cfg, rank, world, 0, None)
                shape = rank_shape(wl, args)
                eng = Engine(shape["plan"], PLAN_ONLY, **shape["engine_kw"])
                try:
                    if eng.flow_reason():
                        continue
                    hi = min(shape["slice"], shape["n"])
                    over = [(c, i, np.ascontiguousarray(v[:hi])) for c, i, v, _ in shape["over"]]
                    spec = eng.jit_spec(shape["seeds"][:hi], over, clock_ptr=8, clock_capacity=shape["clock_cap"],
                                        samples_ptr=8, tick_capacity=shape["ticks"], counts_ptr=8,
                                        draw_capacity=shape["clock_cap"])
                finally:
                    eng.close()
                if spec not in specs:
                    t0 = time.perf_counter()
                    jit.code_object(spec)
                    specs[spec] = f"config {cfg}, {world} GPU(s), rank {rank}"
                    if verbose:
                        print(f"prebuilt af_flow_jit for {specs[spec]} in {time.perf_counter() - t0:.1f} s", flush=True)
    return list(specs)


class RankSweep:
    """Engine, output buffers and per-scenario summaries of this rank's share of the workload."""

    def __init__(self, wl: dict, dev, args) -> None:
        import torch

        from asyncflow_amd import _abi
        from asyncflow_amd.engine import Engine

        self.torch, self._abi, self.args, self.dev = torch, _abi, args, dev
        shape = rank_shape(wl, args)
        self.plan, self.n, self.seeds, self.over = shape["plan"], shape["n"], shape["seeds"], shape["over"]
        self.clock_cap, self.ticks, self.T, self.online = shape["clock_cap"], shape["ticks"], shape["T"], shape["online"]
        self.slice, self.n_slices = shape["slice"], shape["n_slices"]
        plan, n = self.plan, self.n
        self.eng = Engine(plan, dev.index, **shape["engine_kw"])
        self.flow_reason = self.eng.flow_reason()
        m = self.slice
        self.counts = torch.zeros((n, _abi.CNT_SLOTS), dtype=torch.int32, device=dev)
        self.clock = None if self.online else torch.empty((m, self.clock_cap, 2), dtype=torch.float64, device=dev)
        self.o_hist = torch.zeros((m, 1024), dtype=torch.int32, device=dev) if self.online else None
        self.o_rps = torch.zeros((m, max(self.T, 1)), dtype=torch.int32, device=dev) if self.online else None
        self.samples = None if args.no_series else torch.zeros((m, self.ticks, plan.series_pitch), dtype=torch.int32, device=dev)
        # per-scenario summaries (the analyzer step of the path), kept for the whole rank
        self.s_stats = torch.full((n, 8), float("nan"), dtype=torch.float64, device=dev)
        self.s_rps = torch.zeros((n, self.T), dtype=torch.float32, device=dev)
        self.hist_max = {1: 1.024, 2: 0.256, 3: 2.56, 4: 2.56, 5: 25.6, 6: 0.256}[args.config]
        self.s_hist = torch.zeros((n, 256), dtype=torch.int32, device=dev)
        self.s_mean = None if self.samples is None else torch.empty((n, plan.n_series), dtype=torch.float64, device=dev)
        self.s_max = None if self.samples is None else torch.empty((n, plan.n_series), dtype=torch.int32, device=dev)
        self.specialise = not args.generic_kernels

    def _slice_args(self, lo: int, hi: int) -> tuple[np.ndarray, list, dict]:
        seeds = self.seeds[lo:hi]
        over = [(c, i, np.ascontiguousarray(v[lo:hi])) for c, i, v, _ in self.over]
        kw = dict(clock_ptr=self.clock.data_ptr() if self.clock is not None else 0, clock_capacity=self.clock_cap,
                  samples_ptr=self.samples.data_ptr() if self.samples is not None else 0, tick_capacity=self.ticks,
                  counts_ptr=self.counts[lo:hi].data_ptr(), draw_capacity=self.clock_cap)
        if self.online:
            kw.update(online_hist_ptr=self.o_hist.data_ptr(), online_hist_bins=1024, online_hist_max=self.hist_max,
                      online_rps_ptr=self.o_rps.data_ptr(), online_rps_buckets=self.T)
        return seeds, over, kw

    def prepare(self) -> None:
        # plan-specialised build of what the sweep launches (asyncflow_amd/jit.py): the stage-parallel kernel when it runs the
        # plan (one entry point, ~3 s of hipcc once per plan shape and layout, then a cache hit), else the next-event
        # kernels; hand-backs use the library's generic kernels
        self.jit_build_s = 0.0
        if self.specialise:
            seeds, over, kw = self._slice_args(0, min(self.slice, self.n))
            t0 = time.perf_counter()
            self.eng.prepare(seeds, over, **kw)     # hipcc run or cache hit: never inside the timed region
            self.jit_build_s = time.perf_counter() - t0

    def _summary_args(self, lo: int, hi: int) -> dict:
        return dict(stats_ptr=self.s_stats[lo:hi].data_ptr(), rps_ptr=self.s_rps[lo:hi].data_ptr(), rps_buckets=self.T,
                    hist_ptr=self.s_hist[lo:hi].data_ptr(), hist_bins=256, hist_max=self.hist_max,
                    series_mean_ptr=self.s_mean[lo:hi].data_ptr() if self.s_mean is not None else 0,
                    series_max_ptr=self.s_max[lo:hi].data_ptr() if self.s_max is not None else 0)

    def run_slice(self, lo: int, hi: int, summarize: bool = False):
        """af_engine_run over scenarios [lo, hi) of the rank's batch into the (reused) output buffers; `summarize`: the analyzer
        too, in the same call (af_engine_run_summarized: the same results, the analyzer of the stage-parallel kernel's full
        residency rounds beside its last, partial one)."""
        seeds, over, kw = self._slice_args(lo, hi)
        if self.online:
            self.o_hist.zero_()
            self.o_rps.zero_()
        return self.eng.run(seeds, over, specialise=self.specialise, summary=self._summary_args(lo, hi) if summarize else None, **kw)

    def step(self) -> dict:
        """One pass over the rank's batch; returns the engine's own timings summed over the slices."""
        acc = {"kernel_ms": 0.0, "pregen_ms": 0.0, "summary_ms": 0.0, "shared": 0, "jit": 0, "flow_ms": 0.0, "flow_scen": 0,
               "flow_fallback": [0, 0, 0, 0, 0], "jit_fallbacks": 0, "summary_beside_ms": 0.0, "summary_overlapped": 0}
        fused = not self.online and not getattr(self.args, "separate_summary", False)
        for lo in range(0, self.n, self.slice):
            hi = min(self.n, lo + self.slice)
            st = self.run_slice(lo, hi, summarize=fused)
            acc["kernel_ms"] += float(st.kernel_ms)
            acc["pregen_ms"] += float(st.pregen_ms)
            acc["shared"] += int(st.shared_instant_scenarios)
            acc["jit"] += int(st.specialised_launches)
            acc["jit_fallbacks"] += int(st.jit_fallbacks)
            acc["flow_ms"] += float(st.flow_kernel_ms)
            acc["flow_scen"] += int(st.flow_scenarios)
            for k, name in enumerate(("flow_fallback", "flow_fallback_tie", "flow_fallback_list", "flow_fallback_ring", "flow_fallback_ram")):
                acc["flow_fallback"][k] += int(getattr(st, name))
            self.run_stats = {"flow_list_entries": int(st.flow_list_entries), "flow_ring_rows": int(st.flow_ring_rows),
                              "flow_lds_bytes": int(st.flow_lds_bytes), "state_in_lds": int(st.state_in_lds),
                              "lds_bytes_per_wave": int(st.lds_bytes_per_wave), "lanes_per_wave": int(st.lanes_per_wave),
                              "waves": int(st.waves), "request_capacity": int(st.request_capacity),
                              "state_bytes_per_scenario": int(st.state_bytes_per_scenario), "draw_bytes": int(st.draw_bytes),
                              "pregen_group": int(st.pregen_group)}
            if self.online:      # the kernel-side summary IS the analyzer step of this mode
                self.last_stats = st
                continue
            if fused:            # run + analyzer were ONE call
                acc["summary_ms"] += float(st.summary_ms)
                acc["summary_beside_ms"] += float(st.summary_beside_ms)
                acc["summary_overlapped"] += int(st.summary_overlapped)
                self.last_stats = st
                continue
            st = self.eng.summarize(hi - lo, clock_ptr=self.clock.data_ptr(), clock_capacity=self.clock_cap,
                                    samples_ptr=self.samples.data_ptr() if self.samples is not None else 0,
                                    tick_capacity=self.ticks, counts_ptr=self.counts[lo:hi].data_ptr(),
                                    stats_ptr=self.s_stats[lo:hi].data_ptr(), rps_ptr=self.s_rps[lo:hi].data_ptr(),
                                    rps_buckets=self.T, hist_ptr=self.s_hist[lo:hi].data_ptr(), hist_bins=256,
                                    hist_max=self.hist_max,
                                    series_mean_ptr=self.s_mean[lo:hi].data_ptr() if self.s_mean is not None else 0,
                                    series_max_ptr=self.s_max[lo:hi].data_ptr() if self.s_max is not None else 0)
            acc["summary_ms"] += float(st.summary_ms)
            self.last_stats = st
        return acc


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` outside torchrun: re-execute under torch.distributed.run, one rank per GPU."""
    env = dict(os.environ, AF_BENCH_SPAWNED="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(Path(__file__).resolve()), *sys.argv[1:]]
    return subprocess.run(cmd, env=env, check=False).returncode


# --------------------------------------------------------------------------- #
def selftest_cpu(args, rank: int, world: int) -> int:
    """Launcher / sharding / gather path on CPU (gloo), no engine: every rank fabricates the summary
    rows of ITS scenarios (a pure function of the seed), the ranks all-gather them, rank 0 checks that
    the stated total arrived exactly once and prints the one JSON line."""
    import torch
    import torch.distributed as dist

    from asyncflow_amd.distributed import gather_summaries

    if world > 1:
        dist.init_process_group(backend="gloo")
    wl = build_workload(args.config, rank, world, args.scenarios, args.horizon)
    seeds = wl["seeds"].astype(np.float64)
    users = wl["columns"].get("rqs_input.avg_active_users.mean", np.full(wl["n"], 400.0))
    local = torch.tensor(np.stack([seeds, users, np.full(wl["n"], float(rank))], axis=1), dtype=torch.float64)
    # the bench's own scheme: shard sizes computed locally (a pure function of config and world), every rank's scalars
    # in one extra row of its payload, ONE all-gather
    sizes = [wl["n"] if r == rank else build_workload(args.config, r, world, args.scenarios, args.horizon)["n"] for r in range(world)]
    rows = max(sizes) + 1
    padded = torch.zeros((rows, 3), dtype=torch.float64)
    padded[: wl["n"]] = local
    padded[rows - 1] = torch.tensor([float(wl["n"]), float(users.sum()), float(rank)], dtype=torch.float64)
    t0 = time.perf_counter()
    out = gather_summaries(padded, [rows] * world if world > 1 else None)
    gather_ms = (time.perf_counter() - t0) * 1e3
    per_rank = out[torch.arange(world) * rows + rows - 1]
    assert [int(x) for x in per_rank[:, 0].tolist()] == sizes and [int(x) for x in per_rank[:, 2].tolist()] == list(range(world))
    full = torch.cat([out[r * rows: r * rows + sizes[r]] for r in range(world)], dim=0)
    loads = [per_rank[:, 1]]
    if rank == 0:
        got = np.sort(full[:, 0].numpy().astype(np.uint64))
        total = int(full.shape[0])
        line = {"selftest": True, "metric": "launcher selftest (no engine)", "value": float(total), "unit": "scenarios",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "scaling": wl["scaling"],
                "config": {"workload": wl["label"]}, "scenarios_total": total, "unique_seeds": int(np.unique(got).size),
                "scenarios_per_rank": sizes, "gather_ms": gather_ms, "collectives": 1 if world > 1 else 0,
                "load_per_rank": [float(x) for x in torch.cat(loads).tolist()]}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


def make_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=[1, 2, 3, 4, 5, 6], help="BASELINE.json config (see module docstring; 6 = general servers, not a BASELINE config)")
    ap.add_argument("--scenarios", "--replicas", type=int, default=0, dest="scenarios",
                    help="scenarios per GPU (configs 1-3) or in total (configs 4, 5); 0 = the BASELINE size")
    ap.add_argument("--horizon", type=int, default=0, help="simulated seconds (0 = the BASELINE horizon)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-check", action="store_true", help="skip the post-run oracle comparison of >= 16 benched scenarios (every slice of the step)")
    ap.add_argument("--no-diagnostics", action="store_true", help="skip the post-run end-to-end / residency-tail measurements")
    ap.add_argument("--no-series", action="store_true", help="do not store the sampled series")
    ap.add_argument("--lanes", type=int, default=0, help="scenario lanes per wave of the sequential kernel (0 = engine default)")
    ap.add_argument("--global-state", action="store_true", help="keep per-scenario state in HBM")
    ap.add_argument("--generic-kernels", action="store_true",
                    help="do not build plan-specialised kernels (asyncflow_amd/jit.py); use the library's generic ones")
    ap.add_argument("--expect-shared-instants", action="store_true",
                    help="start with the kernel variant that has the SimPy-order path for shared instants")
    ap.add_argument("--online-summary", action="store_true",
                    help="DIAGNOSTIC mode: no per-request clock and no sampled series; the kernel keeps a 1024-bin latency "
                         "histogram and the 1-s completion counts per scenario (19 KB instead of 1.9 MB per LB-2 scenario), "
                         "so that sweeps far beyond the BASELINE sizes fit (e.g. --scenarios 131072)")
    ap.add_argument("--no-flow", action="store_true", help="next-event kernels only (no stage-parallel kernel)")
    ap.add_argument("--separate-summary", action="store_true",
                    help="af_engine_run, then af_engine_summarize (rounds 2-5) instead of the one call af_engine_run_summarized, which "
                         "hides the analyzer of the stage-parallel kernel's full residency rounds beside its last, partial one")
    ap.add_argument("--flow-list-entries", type=int, default=0, choices=[0, 64, 128, 256])
    ap.add_argument("--flow-ring-rows", type=int, default=0, help="rows of the LDS tick ring (0 = auto, -1 = keep the differences in HBM)")
    ap.add_argument("--hbm-budget-gb", type=float, default=96.0, help="HBM for the output buffers of one slice")
    ap.add_argument("--selftest-cpu", action="store_true", help="launcher / sharding / gather path on CPU (gloo), no engine")
    return ap


def main() -> int:  # noqa: C901, PLR0912, PLR0915
    ap = make_parser()
    args = ap.parse_args()
    args.horizon = args.horizon or None

    under_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not under_launcher:
        return spawn_ranks(args.gpus)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.selftest_cpu:
        return selftest_cpu(args, rank, world)

    wl = build_workload(args.config, rank, world, args.scenarios, args.horizon)
    base = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from asyncflow_amd.workloads import BASELINE_SEED_BASE

        base = cpu_baseline(wl["payload"], BASELINE_SEED_BASE[args.config], wl["label"])  # before HIP is initialised (fork-safe)

    import torch

    from asyncflow_amd import _abi

    if not torch.cuda.is_available():
        print("bench.py needs an MI355X: the engine has no CPU fallback", file=sys.stderr)
        if rank == 0:
            print(json.dumps(error_line("no GPU visible (torch.cuda.is_available() is False): the engine has no CPU fallback")), flush=True)
        return 2
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("AF_BENCH_FORCE_DIST") == "1":  # the env flag exercises the RCCL path on 1 GPU
        import torch.distributed as dist  # type: ignore[no-redef]

        dist.init_process_group(backend="nccl", device_id=dev)

    sw = RankSweep(wl, dev, args)
    sw.prepare()

    def barrier() -> None:
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        sw.step()
    barrier()
    t0 = time.perf_counter()
    accs = []
    for _ in range(args.steps):
        accs.append(sw.step())
    barrier()
    elapsed = time.perf_counter() - t0
    n, plan, rs = sw.n, sw.plan, sw.run_stats
    flow_on = accs[-1]["flow_scen"] > 0

    c = sw.counts.cpu().numpy().view(np.uint32)
    flags = int(np.bitwise_or.reduce(c[:, _abi.CNT_FLAGS]))
    if flags & _abi.FATAL_FLAGS:
        print(f"capacity overflow flags={flags:#x}: result invalid", file=sys.stderr)
        return 3
    events_rank = float(c[:, _abi.CNT_EVENTS].astype(np.float64).sum())
    completed_rank = float(c[:, _abi.CNT_COMPLETED].astype(np.float64).sum())
    ticks_rank = float(c[:, _abi.CNT_TICKS].astype(np.float64).sum())
    k_ms = float(np.mean([a["kernel_ms"] for a in accs]))
    summary_ms = float(np.mean([a["summary_ms"] for a in accs]))

    # ---- the single collective: gather per-scenario summaries (after the timed region)
    gather_ms = 0.0
    gather_path = None
    stats_all, hist_all = sw.s_stats, sw.s_hist.to(torch.float32)
    kernel_ms_ranks = [k_ms]
    per_rank_line = None
    n_total = n
    gather_fallback = False
    if dist is not None:
        from asyncflow_amd.distributed import gather_summaries

        # ONE collective: every rank's shard size is a pure function of (config, world) -- no size exchange --, and the
        # rank's own scalars (scenarios, elapsed, events, kernel ms) ride in one extra row of the stats array instead of
        # three all-reduces behind the gather (DESIGN says ONE)
        sizes = [n if r == rank else build_workload(args.config, r, world, args.scenarios, args.horizon)["n"] for r in range(world)]
        n_max = max(sizes)
        rows = n_max + 1
        flow_ms_rank = float(np.mean([a["flow_ms"] for a in accs]))
        rank_row = torch.tensor([[float(n), elapsed, events_rank, k_ms, float(rank), flow_ms_rank, 0.0, 0.0]], dtype=torch.float64, device=dev)
        pad = lambda t: torch.cat([t, torch.zeros((rows - t.shape[0], *t.shape[1:]), dtype=t.dtype, device=dev)], dim=0)  # noqa: E731
        stats_x = pad(sw.s_stats)
        stats_x[n_max] = rank_row[0]
        gather_path = "af_engine_gather (RCCL through the C ABI, one grouped all-gather on the engine's stream)"
        torch.cuda.synchronize(dev)
        try:
            from asyncflow_amd.distributed import EngineComm, gather_engine_summaries

            comm = EngineComm(rank, world, local_rank)
            # what RCCL itself says about the communicator the collective runs on (ncclCommCount / ncclCommUserRank): rides in
            # the gathered rank rows, so that the line of an N-GPU run certifies that RCCL saw N ranks
            stats_x[n_max, 6], stats_x[n_max, 7] = (float(v) for v in comm.count())
            t2 = time.perf_counter()                       # the communicator's setup is not part of the collective
            got = gather_engine_summaries(sw.eng, comm, {"stats": stats_x, "rps": sw.s_rps, "hist": sw.s_hist}, rows)
            torch.cuda.synchronize(dev)
            gather_ms = (time.perf_counter() - t2) * 1e3
            comm.close()
            stats_g, hist_g = got["stats"], got["hist"]
        except Exception as exc:  # noqa: BLE001 - the bench line must survive a broken RCCL install: same collective through torch, LOUDLY
            gather_fallback = True
            gather_path = f"FALLBACK torch.distributed.all_gather_into_tensor (af_engine_gather failed: {type(exc).__name__}: {exc})"
            print(f"[bench] rank {rank}: {gather_path}", file=sys.stderr, flush=True)
            packed = torch.cat([stats_x, pad(sw.s_rps).to(torch.float64), pad(sw.s_hist).to(torch.float64)], dim=1).contiguous()
            torch.cuda.synchronize(dev)
            t2 = time.perf_counter()
            out = gather_summaries(packed, [rows] * world)          # ONE all_gather over xGMI (RCCL)
            torch.cuda.synchronize(dev)
            gather_ms = (time.perf_counter() - t2) * 1e3
            stats_g, hist_g = out[:, :8], out[:, 8 + sw.T:]
        per_rank = stats_g[torch.arange(world, device=dev) * rows + n_max].cpu().numpy()      # the ranks' scalar rows
        assert [int(x) for x in per_rank[:, 0]] == sizes and [int(x) for x in per_rank[:, 4]] == list(range(world)), "gather order"
        keep = torch.cat([torch.arange(r * rows, r * rows + sizes[r], device=dev) for r in range(world)])
        stats_all = stats_g.index_select(0, keep)
        hist_all = hist_g.index_select(0, keep).to(torch.float32)
        n_total = int(stats_all.shape[0])
        elapsed, events_total = float(per_rank[:, 1].max()), float(per_rank[:, 2].sum())
        kernel_ms_ranks = [float(per_rank[:, 3].min()), float(per_rank[:, 3].max())]
        per_rank_line = {"scenarios": [int(x) for x in per_rank[:, 0]], "elapsed_s": [float(x) for x in per_rank[:, 1]],
                         "request_events_per_step": [float(x) for x in per_rank[:, 2]], "kernel_ms": [float(x) for x in per_rank[:, 3]],
                         "flow_kernel_ms": [float(x) for x in per_rank[:, 5]],
                         "rccl_comm_count": None if gather_fallback else [int(x) for x in per_rank[:, 6]],
                         "rccl_comm_user_rank": None if gather_fallback else [int(x) for x in per_rank[:, 7]]}
        if not gather_fallback:
            assert per_rank_line["rccl_comm_count"] == [world] * world and per_rank_line["rccl_comm_user_rank"] == list(range(world)), \
                f"RCCL saw {per_rank_line['rccl_comm_count']} ranks, the launcher {world}"
    else:
        events_total = events_rank

    # pooled p95 over every scenario of the job, from the gathered (or local) 256-bin histograms
    pooled = hist_all.sum(dim=0).double().cpu().numpy()
    cdf = np.cumsum(pooled) / max(pooled.sum(), 1.0)
    pooled_p95_ms = float((np.searchsorted(cdf, 0.95) + 1) * sw.hist_max / 256 * 1e3)

    # ---- diagnostics after the timed region (rank 0, single GPU): end-to-end wall of one sweep, residency tail
    e2e = tail = parity = None
    if rank == 0 and not args.no_parity_check:      # (first: the diagnostics below reuse the output buffers)
        try:
            parity = parity_spot_check(sw)
        except Exception as exc:  # noqa: BLE001 - a broken checker build must not eat the measured line; it says so
            parity = {"scenarios": 0, "ok": None, "why": f"{type(exc).__name__}: {exc}"}
    if rank == 0 and world == 1 and not args.no_diagnostics:
        e2e = end_to_end_sweep(sw, wl, dev)
        if flow_on:
            tail = residency_tail(sw, float(np.mean([a["flow_ms"] for a in accs])))
        c = sw.counts.cpu().numpy().view(np.uint32)     # (the diagnostics ran the same scenarios again: same counts)

    if rank == 0:
        total_events = events_total * args.steps
        alg_bytes = algorithmic_bytes(c, plan.n_series, int(rs["lds_bytes_per_wave"] - rs["lanes_per_wave"] * rs["state_bytes_per_scenario"])
                                      if rs["state_in_lds"] and not flow_on else 0)
        dom_ms = float(np.mean([a["flow_ms"] for a in accs])) if flow_on else k_ms      # the dominant kernel's own duration
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9
        generated_rank = float(c[:, _abi.CNT_GENERATED].astype(np.float64).sum())
        reads = 8.0 * (generated_rank + n) if flow_on else 8.0 * (events_rank + generated_rank)
        io_bytes = reads + (16.0 * completed_rank if sw.clock is not None else 0.0) + (4.0 * plan.series_pitch * ticks_rank if sw.samples is not None else 0.0) + 32.0 * n
        sa = stats_all.double().cpu().numpy()
        out_bytes = 16.0 * completed_rank + (4.0 * plan.series_pitch * ticks_rank if sw.samples is not None else 0.0)
        line = {
            "metric": "simulated request-events/sec (2-server LB scenario, 10k replicas/GPU)" if args.config == 2
                      else f"simulated request-events/sec (BASELINE config {args.config})",
            "value": total_events / elapsed,
            "unit": "request-events/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": wl["scaling"],
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": wl["label"] + (", DIAGNOSTIC: kernel-side summary only (no rqs_clock, no sampled series)" if sw.online
                                           else ", full outputs" + (" without sampled series" if args.no_series else "")),
                "diagnostic": bool(sw.online),
                "baseline_config": args.config,
                "scenarios_rank0": n,
                "scenarios_total": n_total,
                "slices_per_step": sw.n_slices,
                "parallelism": f"scenario-sharded x{world}, no data-path collective",
                "kernel": "af_flow_kernel (stage-parallel, one wave per scenario)" if flow_on else "af_des_kernel (next-event, one scenario per lane)",
                "flow": {"scenarios": accs[-1]["flow_scen"], "list_entries": rs["flow_list_entries"], "ring_rows": rs["flow_ring_rows"],
                         "lds_bytes_per_wave": rs["flow_lds_bytes"],
                         "handed_back": dict(zip(("total", "tie", "list", "ring", "ram"), accs[-1]["flow_fallback"])),
                         "plan_specialised_kernel": bool(flow_on and accs[-1]["jit"]), "jit_fallbacks": int(accs[-1]["jit_fallbacks"]),
                         "jit_build_or_cache_load_s": sw.jit_build_s,
                         "not_used_because": sw.flow_reason or None},
                "next_event": {"state": "LDS" if rs["state_in_lds"] else "HBM", "request_capacity": rs["request_capacity"],
                               "lds_bytes_per_wave": rs["lds_bytes_per_wave"], "lanes_per_wave": rs["lanes_per_wave"],
                               "waves": rs["waves"], "shared_instant_scenarios": int(accs[-1]["shared"]),
                               "plan_specialised_kernels": bool(accs[-1]["jit"]) and not flow_on, "jit_fallbacks": int(accs[-1]["jit_fallbacks"])},
            },
            "events_per_step": events_total,
            "per_gpu_value": total_events / elapsed / world,
            "sweep_wall_s_per_10k": elapsed / args.steps * (10_000 / max(n, 1)),
            # lowering + H2D + kernels + summary D2H of ONE sweep (measured once after the timed region;
            # the one-off hipcc build of the plan-specialised kernel is listed beside it, never inside)
            "sweep_wall_s_per_10k_end_to_end": None if e2e is None else e2e["total_s"] * (10_000 / max(n, 1)),
            "end_to_end": e2e,
            "kernel_ms": k_ms,
            "kernel_ms_ranks_min_max": kernel_ms_ranks,
            "pregen_ms": float(np.mean([a["pregen_ms"] for a in accs])),
            "flow_kernel_ms": float(np.mean([a["flow_ms"] for a in accs])),
            "draw_bytes": rs["draw_bytes"],
            "pregen_group": rs.get("pregen_group"),      # scenarios per workgroup of af_arrival_groups; 0: the row kernel (af_pregen_arrivals_rows)
            "summary_ms": summary_ms,
            "summary": {
                "kernels": "af_summary_kernel + af_series_kernel (inside the timed step)",
                "ms": summary_ms,
                # af_engine_run_summarized: `ms` is what ran AFTER the simulation kernels; `beside_ms` the analyzer kernels that ran on
                # the second stream beside the stage-parallel kernel's last residency round, over `overlapped_scenarios` scenarios
                "call": "af_engine_run, then af_engine_summarize" if args.separate_summary else "af_engine_run_summarized (one call)",
                "beside_ms": float(np.mean([a["summary_beside_ms"] for a in accs])),
                "overlapped_scenarios": int(accs[-1]["summary_overlapped"]),
                # numpy's two-pass variance needs the finished mean: TWO reads of every rqs_clock row (round 6: mean / std_dev bit-equal
                # to numpy's), one of every sample word
                "algorithmic_bytes": out_bytes + 16.0 * completed_rank,
                "achieved_GBps": (out_bytes + 16.0 * completed_rank) / max(summary_ms + float(np.mean([a["summary_beside_ms"] for a in accs])), 1e-9) / 1e6,
            },
            "gather_ms": gather_ms,
            "gather_path": gather_path,
            "gather_fallback": gather_fallback,
            "world_size_launcher": world,
            "rccl_ranks": None if per_rank_line is None or per_rank_line["rccl_comm_count"] is None else per_rank_line["rccl_comm_count"][0],
            "per_rank": per_rank_line,
            "collectives": None if dist is None else "barriers around the timed region + ONE grouped all-gather after it (shard sizes are "
                                                     "computed locally, per-rank scalars ride in the gathered stats array)",
            "p95_ms_mean": None if sw.online else float(np.nanmean(sa[:, 4]) * 1e3),
            "p95_ms_pooled_hist": None if sw.online else pooled_p95_ms,
            "p50_ms_mean": None if sw.online else float(np.nanmean(sa[:, 2]) * 1e3),
            "roofline": {
                # The dominant kernel against the HBM roofline, by the bytes it cannot avoid: the arrival times it reads and every
                # output word it writes (achieved = those bytes / the kernel's own HIP-event time).  The kernel keeps ALL per-scenario
                # state in LDS, so this fraction is small by construction and `binding` names what limits the kernel instead.
                "bound": "hbm",
                "achieved": io_bytes / (dom_ms * 1e-3) / 1e9,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": io_bytes / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "traffic": None,
                "algorithmic_bytes_per_step": io_bytes,
                "algorithmic_bytes_are": "compulsory I/O of the dominant kernel per step: 8 B per arrival time read (generated + 1 per "
                                         "scenario) + 16 B per completed request + 4 B x series pitch x ticks + 32 B of counts"
                                         if flow_on else
                                         "compulsory I/O of the dominant kernel per step: 8 B per pre-generated draw consumed + 16 B per "
                                         "completed request + 4 B x series pitch x ticks + 32 B of counts",
                "launches_per_step": sw.n_slices,
                "bytes_per_event": io_bytes / max(events_rank, 1.0),
                "bound_actual": "instruction issue (VALU + scalar + LDS issue slots of the SIMDs), see `binding`",
                # SURVEY 8d's model prices 80 B of per-event state traffic in HBM (north_star's "SoA in HBM"); this kernel has none
                "model_frac_void": {"frac": achieved / HBM_PEAK_GBS, "achieved_GBps": achieved, "algorithmic_bytes_per_step": alg_bytes,
                                    "bytes_per_event": alg_bytes / max(events_rank, 1.0),
                                    "why_void": "SURVEY 8d: B = 80 N_events + 16 N_completed + 4 n_series N_ticks prices per-event state "
                                                "traffic in HBM; the kernel keeps that state in LDS, so the model's bytes never move and "
                                                "the quotient can exceed 1 -- it is not an HBM fraction"},
                "kernel": ("af_flow_jit (plan-specialised build of af_flow_kernel)" if accs[-1]["jit"] else "af_flow_kernel") if flow_on else
                          "af_jit_lean (plan-specialised build of af_des_kernel)" if accs[-1]["jit"] else "af_des_kernel",
                "kernel_ms": dom_ms,
                "kernel_ms_is": "HIP events on the engine's stream around the kernel's launches of one step, averaged over the timed steps",
                "residency_tail": tail,
            },
        }
        attach_binding(line["roofline"], dom_ms, sw.n_slices, args, n, wl)
        if isinstance(line["roofline"].get("binding"), dict) and line["roofline"]["binding"].get("resource"):
            res_name = str(line["roofline"]["binding"]["resource"])
            line["roofline"]["bound_actual"] = ("instruction issue: the VALU is busy most of the launch (`binding.frac`)" if res_name == "valu_issue"
                                                else res_name) + "; see `binding`"
        if base is not None:
            line["cpu_baseline"] = base
        if parity is not None:
            line["parity_spot_check"] = parity
        try:      # (RCCL prints a version banner through C stdio, block-buffered on a pipe: out with it BEFORE the one JSON line)
            import ctypes

            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(line), flush=True)
    sw.eng.close()
    if dist is not None:
        dist.destroy_process_group()
    return 0


def error_line(message: str) -> dict:
    """The ONE JSON line of a run that could not measure anything: the driver's record then names the reason
    instead of holding `parsed: null`."""
    return {"metric": "simulated request-events/sec", "value": None, "unit": "request-events/s", "error": message[-2000:],
            "kernel_sources_sha1": kernel_sources_sha1()}


if __name__ == "__main__":
    try:
        rc = main()
    except SystemExit:
        raise
    except BaseException as exc:  # noqa: BLE001 - whatever it is, the driver gets one parseable line
        import traceback

        traceback.print_exc()
        if int(os.environ.get("RANK", "0")) == 0:
            print(json.dumps(error_line(f"{type(exc).__name__}: {exc}")), flush=True)
        rc = 1
    raise SystemExit(rc)
