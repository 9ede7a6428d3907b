"""GPU fuzz of the round-4 / round-5 additions to the stage-parallel kernel (FEAT_CHAIN, FEAT_GENSRV with shared instants and the
round-at-once solver, both together; servers in front of the LB, deeper tiers, wider fan-outs):
every scenario of every payload against the next-event kernels (counts, every (start, finish) pair, every sample) and two of
them against the oracle.  Prints one JSON line of tallies (profiles/r04/gpu_fuzz_f3.json).  A mismatch raises."""
import json
import random
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from asyncflow_amd import _abi  # noqa: E402
from asyncflow_amd.plan import lower  # noqa: E402
from asyncflow_amd.runner import SimulationRunner  # noqa: E402
from oracle import oracle_lib as ol  # noqa: E402
from oracle.scenarios import deep_chain, flow_payload, gateway_lb, random_payload, server_tiers, tie_storm, wide_fanout  # noqa: E402

def _gateway(k: int) -> dict:
    rng = random.Random(99000 + k)
    return gateway_lb(front=rng.choice((1, 2)), algo=rng.choice(("round_robin", "least_connection")), users=rng.choice((60, 150, 300)),
                      horizon=10, general=rng.random() < 0.4, backend=rng.random() < 0.5, spike=rng.random() < 0.5)


def _deep(k: int) -> dict:
    rng = random.Random(99500 + k)
    return deep_chain(rng.choice((4, 5)), users=rng.choice((60, 150)), horizon=10, fan=rng.random() < 0.6)


def _wide(k: int) -> dict:
    rng = random.Random(99700 + k)
    p = wide_fanout(rng.choice((13, 14, 16)), "round_robin", horizon=12, users=rng.choice((60, 100)))   # (its events reach to 9.5 s)
    for s in p["topology_graph"]["nodes"]["servers"]:
        s["endpoints"] = s["endpoints"][:1]
    return p


def _wide_lc(k: int) -> dict:
    rng = random.Random(99900 + k)
    p = wide_fanout(rng.choice((9, 11, 12, 14, 16)), "least_connection", horizon=12, users=rng.choice((60, 100, 200)))
    if rng.random() < 0.7:   # (else: both endpoints -- general servers behind least connections)
        for s in p["topology_graph"]["nodes"]["servers"]:
            s["endpoints"] = s["endpoints"][:1]
    return p


families = {
    # round 6: 9 .. 16 servers behind a least-connections LB (Flow::lb_pick_lc_n<16>)
    "9 .. 16 servers behind a least-connections LB": _wide_lc,
    # round 5: servers in front of the LB, four / five server levels, 13 .. 16 servers behind a round-robin LB
    "servers in front of the LB": _gateway,
    "four / five server levels": _deep,
    "13 .. 16 servers behind a round-robin LB": _wide,
    "server tiers (tandem)": lambda k: server_tiers(random.Random(95000 + k), horizon=12),
    "server tiers (general servers)": lambda k: server_tiers(random.Random(96000 + k), horizon=12, general=True),
    "random topologies (general servers)": lambda k: random_payload(random.Random(97000 + k), horizon=8),
    "tie storms": lambda k: tie_storm(random.Random(98000 + k), horizon=8),
    # rounds 2-3: the feed-forward range itself (idle to saturated tandem servers, every latency law, dyadic step times, tight RAM,
    # spikes, outages, least connections, both generators)
    "feed-forward payloads (tandem servers)": lambda k: flow_payload(random.Random(94000 + k), horizon=6),
}


def run(n_payloads: int = 40, k0: int = 0, only: str | None = None, oracle_every: bool = False) -> dict:
    """`n_payloads` payloads per family from index `k0` on (a different range is a different set of payloads), eight scenarios
    each; `only`: the families whose name contains it; `oracle_every`: all eight scenarios against the oracle (default: two).
    A mismatch raises."""
    fams = {n: f for n, f in families.items() if only is None or only in n}
    out = {}
    for name, make in fams.items():
        t = {"payloads": 0, "scenarios": 0, "on_flow_kernel": 0, "handed_back_first": 0, "to_next_event": 0, "oracle_checks": 0, "not_in_range": 0, "overflow_raised": 0,
             # where the default run's kernel time went (VERDICT r5 item 7 asks for the split): stage-parallel kernel (both launches) / next-event kernels
             "flow_kernel_ms": 0.0, "next_event_kernel_ms": 0.0}
        for k in range(k0, k0 + n_payloads):
            payload = make(k)
            seeds = np.arange(8, dtype=np.uint64) + 1000 * k + 7
            try:
                res = SimulationRunner(simulation_input=payload, seeds=seeds, on_negative_delay="flag").run()
                ref = SimulationRunner(simulation_input=payload, seeds=seeds, flow=False, on_negative_delay="flag").run()
            except OverflowError:   # a pool at the engine's maximum: reported, never silent (runner.py)
                t["overflow_raised"] += 1
                continue
            st = res.engine_stats
            t["payloads"] += 1
            t["scenarios"] += 8
            if res.flow_reason:
                t["not_in_range"] += 1
            t["on_flow_kernel"] += int(st.flow_scenarios)
            t["handed_back_first"] += int(st.flow_fallback)
            t["to_next_event"] += int(st.flow_to_next_event)
            t["flow_kernel_ms"] += float(st.flow_kernel_ms)
            t["next_event_kernel_ms"] += float(st.kernel_ms) - float(st.flow_kernel_ms)
            assert np.array_equal(res.counts[:, :6], ref.counts[:, :6]), (name, k)
            assert np.array_equal(res.counts[:, _abi.CNT_MARKS], ref.counts[:, _abi.CNT_MARKS]), (name, k)
            for i in range(8):
                assert np.array_equal(res[i].rqs_clock.view(np.uint64), ref[i].rqs_clock.view(np.uint64)), (name, k, i)
                assert np.array_equal(res[i]._samples, ref[i]._samples), (name, k, i)  # noqa: SLF001
            plan = lower(payload)
            for i in (range(8) if oracle_every else (0, 7)):
                want = ol.simulate(plan, int(seeds[i]))
                assert np.array_equal(res[i].counts[:5].astype(np.uint64), want.counts[:5]), (name, k, i)
                assert np.array_equal(res[i].rqs_clock.view(np.uint64), want.clock.view(np.uint64)), (name, k, i)
                assert np.array_equal(res[i]._samples, want.samples), (name, k, i)  # noqa: SLF001
                t["oracle_checks"] += 1
        out[name] = t
    out["different"] = 0
    return out


if __name__ == "__main__":
    print(json.dumps(run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 0,
                         sys.argv[3] if len(sys.argv) > 3 else None)))
