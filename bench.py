"""Benchmark of the hot path: batched `env.run(until=T)` on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--replicas R]

Workload (BASELINE.json configs[1], the one the metric is quoted on): the
2-server + load-balancer topology of examples/yaml_input/data/two_servers_lb.yml
(400 users x 20 rpm, T = 600 s, 0.05 s sampling), 10 000 seed replicas per GPU,
scenario i uses Philox key 0x5EED0000 + i, full-fidelity outputs (every
(start, finish) pair and all 12 sampled series written to HBM).

A "step" is ONE pass of the hot path over that batch: af_engine_run() = seed
upload + the HIP next-event kernel + stream sync, inputs/outputs resident in HBM.
K steps are timed between barrier + torch.cuda.synchronize() on both sides, MAX
over ranks; rank 0 prints ONE JSON line.  value = request-events simulated by all
ranks / that time (request-event = one timed state transition of a request:
arrival, edge delivery, CPU-step end, I/O-step end).

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL), scenarios
sharded by rank with NO data-path collective (weak scaling: R replicas per GPU);
ONE all_gather of the per-scenario summaries over xGMI after the timed region
(reported as gather_ms).
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6290 GB/s measured-achievable
HBM_ACHIEVABLE_GBS = 6290.0
SEED_BASE = 0x5EED0000


def lb2_payload(horizon: int = 600) -> dict:
    from oracle.scenarios import lb_two_servers  # pure dict builder (the YAML's values), no oracle code

    return lb_two_servers(horizon=horizon)


def algorithmic_bytes(counts: np.ndarray, n_series: int, plan_bytes: int) -> float:
    """B = 80 N_events + 16 N_completed + 4 n_series N_ticks + plan_bytes, per scenario."""
    ev = counts[:, 3].astype(np.float64).sum()
    comp = counts[:, 1].astype(np.float64).sum()
    ticks = counts[:, 4].astype(np.float64).sum()
    return 80.0 * ev + 16.0 * comp + 4.0 * n_series * ticks + float(plan_bytes) * counts.shape[0]


# --------------------------------------------------------------------------- #
# CPU baseline: the SimPy-faithful C restatement (oracle/des_oracle.c), "port"  #
# --------------------------------------------------------------------------- #
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


def cpu_baseline(payload: dict, budget_s: float = 12.0) -> dict:
    """Time the oracle on the host cores over a bounded sample of the same workload."""
    import multiprocessing as mp

    from asyncflow_amd.plan import lower
    from oracle import oracle_lib as ol

    ol.build()
    plan = lower(payload)
    t0 = time.perf_counter()
    ol.simulate(plan, SEED_BASE)
    one = max(time.perf_counter() - t0, 1e-3)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # (a run takes ~5x longer with every hardware thread busy than alone: 16 per core ~ 15-20 s of wall)
    per_core = max(2, min(16, int(budget_s / one)))
    jobs = [(payload, [SEED_BASE + c * per_core + k for k in range(per_core)]) for c in range(cores)]
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(cores) as pool:
        parts = pool.map(_cpu_worker, jobs)
    wall = time.perf_counter() - t0
    ev = sum(p[0] for p in parts)
    heap = sum(p[1] for p in parts)
    busy = max(p[2] for p in parts)
    return {
        "value": ev / busy,
        "unit": "request-events/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{cores * per_core} LB-2 replicas (T=600 s) of the same workload, {per_core} per core, "
                  f"C restatement of the reference actors + SimPy heap (oracle/des_oracle.c), "
                  f"{heap / max(ev, 1):.1f} SimPy heap events per request-event, wall {wall:.1f} s",
        "replicas_per_s": cores * per_core / busy,
        "python_reference_events_per_s_per_core": 4.5e4,  # measured in the build container
    }


# --------------------------------------------------------------------------- #
def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--replicas", type=int, default=10_000, help="scenarios per GPU")
    ap.add_argument("--horizon", type=int, default=600)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-series", action="store_true", help="do not store the sampled series")
    ap.add_argument("--lanes", type=int, default=0, help="scenario lanes per wave (0 = engine default)")
    ap.add_argument("--global-state", action="store_true", help="keep per-scenario state in HBM")
    ap.add_argument("--generic-kernels", action="store_true",
                    help="do not build plan-specialised kernels (asyncflow_amd/jit.py); use the library's generic ones")
    ap.add_argument("--expect-shared-instants", action="store_true",
                    help="start with the kernel variant that has the SimPy-order path for shared instants")
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5],
                    help="BASELINE.json config: 2 = 10k LB-2 seed replicas (the metric's config, default); "
                         "3 = users x RTT 100x100 grid; 4 = grid + injected spikes/outages; 5 = 8-server fan-out")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world
    payload = lb2_payload(args.horizon)
    sweep: dict = {}
    label = "two_servers_lb.yml (2 servers + LB, 400 users x 20 rpm"
    if args.config in (3, 4):
        from oracle.scenarios import lb_with_events

        if args.config == 4:
            payload = lb_with_events(users=400, horizon=args.horizon, scale=args.horizon / 600.0)
        label = "LB-2 grid avg_active_users x RTT" + (" + event_inj_lb.yml spikes/outages" if args.config == 4 else "")
    if args.config == 5:
        from oracle.scenarios import fanout8

        payload = fanout8(horizon=args.horizon)
        label = "8-server fan-out, log-normal edges (120 users x 20 rpm"

    base = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        base = cpu_baseline(payload)  # before CUDA is initialised (fork-safe)

    import torch

    from asyncflow_amd import _abi
    from asyncflow_amd.engine import Engine
    from asyncflow_amd.plan import estimate_capacities, lower

    if not torch.cuda.is_available():
        print("bench.py needs an MI355X: the engine has no CPU fallback", file=sys.stderr)
        return 2
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("AF_BENCH_FORCE_DIST") == "1":  # the env flag exercises the RCCL path on 1 GPU
        import torch.distributed as dist  # type: ignore[no-redef]

        dist.init_process_group(backend="nccl", device_id=dev)

    plan = lower(payload)
    n = args.replicas
    seeds = (SEED_BASE + rank * n + np.arange(n, dtype=np.uint64)).astype(np.uint64)
    overrides = []
    users_max = None
    lat_scale = 1.0
    if args.config in (3, 4):
        # users = 10 a (a = 1..100), per-hop latency mean = 0.5 ms b (b = 1..100);
        # scenarios are dealt to lanes by expected load so that a wave holds similar scenarios
        from asyncflow_amd.runner import resolve_sweep

        side = int(round(n ** 0.5))
        n = side * side
        a = np.repeat(np.arange(1, side + 1), side) * (1000.0 / side)
        b = np.tile(np.arange(1, side + 1), side) * (0.05 / side)
        order = np.argsort(-a, kind="stable")
        a, b = a[order], b[order]
        seeds = (0xC0F30000 + rank * n + np.arange(n, dtype=np.uint64)).astype(np.uint64)
        overrides = [(c, i, v) for c, i, v, _ in resolve_sweep(plan, {
            "rqs_input.avg_active_users.mean": a, "topology_graph.edges[*].latency.mean": b}, n)]
        users_max, lat_scale = float(a.max()), float(b.max() / plan.edge_mean.min())
    cap, fifo = estimate_capacities(plan, users_max, lat_scale)
    clock_cap = plan.clock_capacity(users_max)
    ticks = max(plan.tick_count, 1)
    eng = Engine(plan, local_rank, request_capacity=cap, fifo_capacity=fifo, lanes_per_wave=args.lanes,
                 force_global_state=args.global_state, expect_shared_instants=args.expect_shared_instants)
    counts = torch.zeros((n, _abi.CNT_SLOTS), dtype=torch.int32, device=dev)
    clock = torch.empty((n, clock_cap, 2), dtype=torch.float64, device=dev)
    samples = None if args.no_series else torch.zeros((n, ticks, plan.series_pitch), dtype=torch.int32, device=dev)

    # per-scenario summaries (the analyzer step of the path): 8 latency stats, 1-s RPS windows,
    # mean/max of every sampled series -- computed by the HIP analyzer inside the timed step
    T = int(plan.total_time)
    s_stats = torch.empty((n, 8), dtype=torch.float64, device=dev)
    s_rps = torch.empty((n, T), dtype=torch.float32, device=dev)
    # 256-bin latency histogram per scenario (pooled percentiles across replicas / ranks)
    hist_max = {2: 0.256, 3: 2.56, 4: 2.56, 5: 25.6}[args.config]
    s_hist = torch.empty((n, 256), dtype=torch.int32, device=dev)
    s_mean = None if samples is None else torch.empty((n, plan.n_series), dtype=torch.float64, device=dev)
    s_max = None if samples is None else torch.empty((n, plan.n_series), dtype=torch.int32, device=dev)

    run_kw = dict(clock_ptr=clock.data_ptr(), clock_capacity=clock_cap,
                  samples_ptr=samples.data_ptr() if samples is not None else 0, tick_capacity=ticks,
                  counts_ptr=counts.data_ptr(), draw_capacity=clock_cap)
    if not args.generic_kernels:
        eng.prepare(seeds, overrides, **run_kw)     # hipcc run or cache hit: never inside the timed region

    def step():
        eng.run(seeds, overrides, specialise=not args.generic_kernels, **run_kw)
        return eng.summarize(n, clock_ptr=clock.data_ptr(), clock_capacity=clock_cap,
                             samples_ptr=samples.data_ptr() if samples is not None else 0, tick_capacity=ticks,
                             counts_ptr=counts.data_ptr(), stats_ptr=s_stats.data_ptr(), rps_ptr=s_rps.data_ptr(),
                             rps_buckets=T, hist_ptr=s_hist.data_ptr(), hist_bins=256, hist_max=hist_max,
                             series_mean_ptr=s_mean.data_ptr() if s_mean is not None else 0,
                             series_max_ptr=s_max.data_ptr() if s_max is not None else 0)

    def barrier() -> None:
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    kernel_ms = []
    for _ in range(args.steps):
        st = step()
        kernel_ms.append(st.kernel_ms)
    barrier()
    elapsed = time.perf_counter() - t0

    c = counts.cpu().numpy().view(np.uint32)
    flags = int(np.bitwise_or.reduce(c[:, _abi.CNT_FLAGS]))
    if flags & _abi.FATAL_FLAGS:
        print(f"capacity overflow flags={flags:#x}: result invalid", file=sys.stderr)
        return 3
    events_rank = float(c[:, _abi.CNT_EVENTS].astype(np.float64).sum())

    # ---- the single collective: gather per-scenario summaries (after the timed region)
    summ = {"stats": s_stats, "rps": s_rps}
    summary_ms = float(st.summary_ms)
    completed_rank = float(c[:, _abi.CNT_COMPLETED].astype(np.float64).sum())
    ticks_rank = float(c[:, _abi.CNT_TICKS].astype(np.float64).sum())
    gather_ms = 0.0
    stats_all = summ["stats"]
    if dist is not None:
        t2 = time.perf_counter()
        from asyncflow_amd.distributed import gather_summaries

        packed = torch.cat([summ["stats"].to(torch.float32), summ["rps"], s_hist.to(torch.float32)], dim=1).contiguous()
        out = gather_summaries(packed, [n] * world)   # ONE all_gather over xGMI (RCCL)
        torch.cuda.synchronize(dev)
        gather_ms = (time.perf_counter() - t2) * 1e3
        stats_all = out[:, :8]
        tot = torch.tensor([elapsed, events_rank], dtype=torch.float64, device=dev)
        mx = tot.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = tot.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed = float(mx[0])
        events_total = float(sm[1])
    else:
        events_total = events_rank

    # pooled p95 over every scenario of the job, from the gathered (or local) 256-bin histograms
    hist_all = out[:, 8 + T:] if dist is not None else s_hist.to(torch.float32)
    pooled = hist_all.sum(dim=0).double().cpu().numpy()
    cdf = np.cumsum(pooled) / max(pooled.sum(), 1.0)
    pooled_p95_ms = float((np.searchsorted(cdf, 0.95) + 1) * hist_max / 256 * 1e3)

    if rank == 0:
        total_events = events_total * args.steps
        k_ms = float(np.mean(kernel_ms))
        alg_bytes = algorithmic_bytes(c, plan.n_series, int(st.lds_bytes_per_wave - 64 * st.state_bytes_per_scenario)
                                      if st.state_in_lds else 0)
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        sa = stats_all.double().cpu().numpy()
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
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{label}, T={args.horizon} s, dt=0.05 s), "
                            f"{n} scenarios per GPU, full outputs"
                            + (" without sampled series" if args.no_series else ""),
                "scenarios_per_gpu": n,
                "parallelism": f"scenario-sharded x{world}, no data-path collective",
                "state": "LDS" if st.state_in_lds else "HBM",
                "request_capacity": int(st.request_capacity),
                "lds_bytes_per_wave": int(st.lds_bytes_per_wave),
                "lanes_per_wave": int(st.lanes_per_wave),
                "waves": int(st.waves),
                "shared_instant_scenarios": int(st.shared_instant_scenarios),
                "plan_specialised_kernels": bool(st.specialised_launches),
            },
            "events_per_step": events_total,
            "per_gpu_value": total_events / elapsed / world,
            "sweep_wall_s_per_10k": elapsed / args.steps * (10_000 / n),
            "kernel_ms": k_ms,
            "pregen_ms": float(st.pregen_ms),
            "draw_bytes": int(st.draw_bytes),
            "summary_ms": summary_ms,
            "summary": {
                "kernels": "af_summary_kernel + af_series_kernel (inside the timed step)",
                "ms": summary_ms,
                # one read of every rqs_clock row and sample word is the algorithmic minimum
                "algorithmic_bytes": 16.0 * completed_rank + (4.0 * plan.series_pitch * ticks_rank if samples is not None else 0.0),
                "achieved_GBps": (16.0 * completed_rank + (4.0 * plan.series_pitch * ticks_rank if samples is not None else 0.0))
                                 / max(summary_ms, 1e-9) / 1e6,
            },
            "gather_ms": gather_ms,
            "p95_ms_mean": float(np.nanmean(sa[:, 4]) * 1e3),
            "p95_ms_pooled_hist": pooled_p95_ms,
            "p50_ms_mean": float(np.nanmean(sa[:, 2]) * 1e3),
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "frac_of_achievable_6290": achieved / HBM_ACHIEVABLE_GBS,
                "traffic": None,
                "algorithmic_bytes_per_launch": alg_bytes,
                "bytes_per_event": alg_bytes / max(events_rank, 1.0),
                "kernel": "af_jit_lean (plan-specialised build of af_des_kernel)" if st.specialised_launches else "af_des_kernel",
            },
        }
        if base is not None:
            line["cpu_baseline"] = base
            line["gpu_over_cpu_port_all_cores"] = (total_events / elapsed) / base["value"]
        # HBM bytes per launch of the dominant kernel, from the committed PMC passes of THIS
        # command (rocprofv3 cannot run inside the timed bench): profiles/r01/final/traffic.json
        tpath = ROOT / "profiles" / "r01" / "final" / "traffic.json"
        if args.config == 2 and n == 10_000 and not args.no_series and args.horizon == 600 and tpath.exists():
            tj = json.loads(tpath.read_text())
            line["roofline"]["traffic"] = tj["bytes_per_launch"]
            line["roofline"]["traffic_source"] = "profiles/r01/final/traffic.json (2 x FETCH_SIZE + WRITE_SIZE, KB=1024 B)"
        print(json.dumps(line))
    eng.close()
    if dist is not None:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
