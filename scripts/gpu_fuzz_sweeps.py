"""GPU fuzz of the sweep columns: a random feed-forward payload, one to four random columns over the paths
`resolve_sweep` accepts, six scenarios -- every scenario against the ORACLE run on the payload a user of the reference would
have built for that point (`write_point`), and the whole batch against the next-event kernels.

    python scripts/gpu_fuzz_sweeps.py [payloads, default 100] [first payload index, default 0]

Prints one JSON line of tallies (`different`: payloads with any difference, the first ten messages in `failures`)."""
import copy
import json
import random
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from asyncflow_amd import _abi  # noqa: E402
from asyncflow_amd.plan import lower  # noqa: E402
from asyncflow_amd.runner import SimulationRunner, write_point  # noqa: E402
from oracle import oracle_lib as ol  # noqa: E402
from oracle.scenarios import flow_payload, lb_with_events  # noqa: E402

N = 6


def columns(payload: dict, rng: random.Random) -> dict[str, np.ndarray]:
    """One to four columns, each a path of the payload with values around what the payload holds."""
    gen = payload["rqs_input"]
    edges = payload["topology_graph"]["edges"]
    servers = payload["topology_graph"]["nodes"]["servers"]
    users = float(gen["avg_active_users"]["mean"])
    rpm = float(gen["avg_request_per_minute_per_user"]["mean"])
    choices = []
    choices.append(("rqs_input.avg_active_users.mean", lambda: max(1.0, round(users * rng.uniform(0.3, 1.6)))))
    choices.append(("rqs_input.avg_request_per_minute_per_user.mean", lambda: max(1.0, round(rpm * rng.uniform(0.5, 1.5)))))
    choices.append(("rqs_input.user_sampling_window", lambda: float(rng.choice((1, 2, 3, 5)))))
    if "variance" in gen["avg_active_users"] and gen["avg_active_users"].get("distribution") == "normal":
        choices.append(("rqs_input.avg_active_users.variance", lambda: round(users * rng.uniform(0.05, 0.4), 3)))
    e = rng.choice(edges)
    mean = float(e["latency"]["mean"])
    choices.append(("topology_graph.edges[*].latency.mean", lambda: mean * rng.uniform(0.5, 3.0)))
    choices.append((f"topology_graph.edges[{e['id']}].latency.mean", lambda: mean * rng.uniform(0.5, 3.0)))
    choices.append((f"topology_graph.edges[{e['id']}].dropout_rate", lambda: rng.choice((0.0, 0.01, 0.05, 0.2))))
    if e["latency"].get("variance") is not None and e["latency"]["distribution"] in ("normal", "log_normal"):
        var = float(e["latency"]["variance"])
        choices.append((f"topology_graph.edges[{e['id']}].latency.variance", lambda: var * rng.uniform(0.5, 2.0)))
    s = rng.choice(servers)
    for j, ep in enumerate(s["endpoints"]):
        for k, st in enumerate(ep["steps"]):
            for f in ("cpu_time", "io_waiting_time"):
                if f in st["step_operation"]:
                    base = float(st["step_operation"][f])
                    choices.append((f"topology_graph.nodes.servers[{s['id']}].endpoints[{j}].steps[{k}].{f}",
                                    lambda base=base: base * rng.uniform(0.5, 1.5)))
    cores, ram = int(s["server_resources"]["cpu_cores"]), int(s["server_resources"]["ram_mb"])
    choices.append((f"topology_graph.nodes.servers[{s['id']}].server_resources.cpu_cores", lambda: float(rng.choice((cores, cores + 1, max(1, cores - 1))))))
    choices.append((f"topology_graph.nodes.servers[{s['id']}].server_resources.ram_mb", lambda: float(rng.choice((ram, ram * 2, ram + 256)))))
    for ev in payload.get("events") or []:
        if "spike_s" in ev["start"] and ev["start"]["spike_s"] is not None:
            sp = float(ev["start"]["spike_s"])
            choices.append((f"events[{ev['event_id']}].start.spike_s", lambda sp=sp: sp * rng.uniform(0.5, 2.0)))
        t0, t1 = float(ev["start"]["t_start"]), float(ev["end"]["t_end"])   # (windows of one target must not overlap: move them inwards only)
        choices.append((f"events[{ev['event_id']}].start.t_start", lambda t0=t0, t1=t1: t0 + (t1 - t0) * rng.uniform(0.0, 0.4)))
        choices.append((f"events[{ev['event_id']}].end.t_end", lambda t0=t0, t1=t1: t1 - (t1 - t0) * rng.uniform(0.0, 0.4)))
    picked = rng.sample(choices, k=min(len(choices), rng.randint(1, 4)))
    cols: dict[str, np.ndarray] = {}
    for key, draw in picked:
        cols[key] = np.array([draw() for _ in range(N)], dtype=np.float64)
    return cols


def run(n_payloads: int = 100, k0: int = 0) -> dict:
    """Tallies of `n_payloads` payloads from index `k0` on; `different`: payloads with any difference."""
    t = {"payloads": 0, "scenarios": 0, "columns": 0, "on_flow_kernel": 0, "to_next_event": 0, "oracle_checks": 0, "overflow_raised": 0, "invalid_points": 0,
         "paths": {}}
    failures: list[str] = []
    for k in range(k0, k0 + n_payloads):
        rng = random.Random(77000 + k)
        payload = flow_payload(rng, horizon=6) if k % 4 else lb_with_events(users=rng.choice((60, 120, 300)), horizon=30, scale=0.05)
        cols = columns(payload, rng)
        seeds = np.arange(N, dtype=np.uint64) + 1000 * k + 11
        try:
            res = SimulationRunner(simulation_input=payload, seeds=seeds, sweep=cols, on_negative_delay="flag").run()
            ref = SimulationRunner(simulation_input=payload, seeds=seeds, sweep=cols, flow=False, on_negative_delay="flag").run()
        except OverflowError as exc:   # a pool at the engine's maximum: reported, never silent
            t["overflow_raised"] += 1
            print(f"payload {k}: OverflowError: {str(exc)[:200]}", file=sys.stderr)
            continue
        except ValueError as exc:   # a point the payload models refuse (e.g. an event window that no longer fits)
            t["invalid_points"] += 1
            print(f"payload {k}: {str(exc)[:200]}", file=sys.stderr)
            continue
        st = res.engine_stats
        t["payloads"] += 1
        t["scenarios"] += N
        t["columns"] += len(cols)
        for key in cols:
            short = key.split("[")[0] + ("[..]" + key.split("]")[-1] if "[" in key else "")
            t["paths"][short] = t["paths"].get(short, 0) + 1
        t["on_flow_kernel"] += int(st.flow_scenarios)
        t["to_next_event"] += int(st.flow_to_next_event)
        try:
            assert np.array_equal(res.counts[:, :6], ref.counts[:, :6]), (k, list(cols))
            assert np.array_equal(res.counts[:, _abi.CNT_MARKS], ref.counts[:, _abi.CNT_MARKS]), (k, list(cols))
            base = lower(payload).payload   # (normalised: what write_point expects)
            for i in range(N):
                assert np.array_equal(res[i].rqs_clock.view(np.uint64), ref[i].rqs_clock.view(np.uint64)), (k, i, list(cols))
                assert np.array_equal(res[i]._samples, ref[i]._samples), (k, i, list(cols))  # noqa: SLF001
                point = copy.deepcopy(base)
                for key, col in cols.items():
                    write_point(point, key, col[i])
                want = ol.simulate(lower(point), int(seeds[i]))
                assert np.array_equal(res[i].counts[:5].astype(np.uint64), want.counts[:5]), (k, i, list(cols), res[i].counts[:5], want.counts[:5])
                assert np.array_equal(res[i].rqs_clock.view(np.uint64), want.clock.view(np.uint64)), (k, i, list(cols))
                assert np.array_equal(res[i]._samples, want.samples), (k, i, list(cols))  # noqa: SLF001
                t["oracle_checks"] += 1
        except AssertionError as exc:   # (keep going: every differing payload is worth knowing)
            failures.append(str(exc)[:400])
            print(f"DIFFERENT payload {k}: {str(exc)[:400]}", file=sys.stderr)
    t["different"] = len(failures)
    t["failures"] = failures[:10]
    return t


if __name__ == "__main__":
    print(json.dumps(run(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 0)))
