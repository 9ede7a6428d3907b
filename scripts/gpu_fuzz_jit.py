"""Fuzz of the plan-specialised kernels (asyncflow_amd/jit.py): random payloads of seven families, each run on its OWN specialised
build of af_flow_jit and on the library's generic instantiation -- counts, every (start, finish) pair and every sample equal, and two
scenarios against the oracle.

    python scripts/gpu_fuzz_jit.py prebuild [payloads, default 60] [first index]   # build container: hipcc, no GPU (SimulationRunner.prebuild)
    python scripts/gpu_fuzz_jit.py run [payloads] [first index]                      # GPU box: ASYNCFLOW_NO_HIPCC=1, the cache travels in-tree

One JSON line: how many payloads ran a specialised kernel (a spec that was not prebuilt -- a second-chance launch, a retry with
larger pools -- falls back to the generic kernel: `jit_fallbacks`), `different`."""
import json
import os
import random
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle.scenarios import deep_chain, flow_payload, gateway_lb, random_payload, server_tiers, tie_storm, wide_fanout  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "run"
n_payloads = int(sys.argv[2]) if len(sys.argv) > 2 else 60
k0 = int(sys.argv[3]) if len(sys.argv) > 3 else 0
N = 8


def make(k: int) -> dict:
    rng = random.Random(66000 + k)
    kind = k % 7
    if kind == 0:
        return random_payload(rng, horizon=8)
    if kind == 1:
        return server_tiers(rng, horizon=10, general=rng.random() < 0.5)
    if kind == 2:
        return tie_storm(rng, horizon=8)
    if kind == 3:
        return gateway_lb(front=rng.choice((1, 2)), algo=rng.choice(("round_robin", "least_connection")), users=rng.choice((60, 300)), horizon=10,
                          general=rng.random() < 0.3, backend=rng.random() < 0.5, spike=rng.random() < 0.5)
    if kind == 4:
        return deep_chain(rng.choice((3, 4, 5)), users=rng.choice((60, 150)), horizon=10, fan=rng.random() < 0.6)
    if kind == 5:
        p = wide_fanout(rng.choice((9, 13, 16)), "round_robin", horizon=12, users=rng.choice((60, 100)))
        for s in p["topology_graph"]["nodes"]["servers"]:
            s["endpoints"] = s["endpoints"][:1]
        return p
    return flow_payload(rng, horizon=6)


def runner_of(k: int, **kw):
    from asyncflow_amd.runner import SimulationRunner

    return SimulationRunner(simulation_input=make(k), seeds=np.arange(N, dtype=np.uint64) + 50 * k + 1, on_negative_delay="flag", **kw)


def prebuild_one(k: int) -> str:
    from asyncflow_amd.engine import EngineUnavailableError

    try:
        return "built" if runner_of(k, specialise=True).prebuild() else "no spec"
    except EngineUnavailableError:
        return "next-event plan"     # (its kernels' shape depends on the device: built where they run)
    except Exception as exc:  # noqa: BLE001 - tallied
        return f"{type(exc).__name__}: {str(exc)[:120]}"


def main() -> None:
    if mode == "prebuild":
        import collections
        import multiprocessing as mp

        with mp.get_context("spawn").Pool(max(1, (os.cpu_count() or 2) - 1)) as pool:
            tally = collections.Counter(pool.map(prebuild_one, range(k0, k0 + n_payloads), chunksize=1))
        print(json.dumps(dict(tally)))
        return

    from asyncflow_amd import _abi  # noqa: E402
    from asyncflow_amd.plan import lower  # noqa: E402
    from oracle import oracle_lib as ol  # noqa: E402

    t = {"payloads": 0, "scenarios": 0, "ran_specialised": 0, "jit_fallbacks": 0, "not_on_flow_kernel": 0, "overflow_raised": 0, "oracle_checks": 0, "generic_by_family": {}}
    failures: list[str] = []
    for k in range(k0, k0 + n_payloads):
        try:
            res = runner_of(k, specialise=True).run()
            gen = runner_of(k, specialise=False).run()
        except OverflowError:
            t["overflow_raised"] += 1
            continue
        st = res.engine_stats
        t["payloads"] += 1
        t["scenarios"] += N
        t["ran_specialised"] += int(st.specialised_launches > 0)
        t["jit_fallbacks"] += int(st.jit_fallbacks)
        if st.specialised_launches == 0:
            t["generic_by_family"][str(k % 7)] = t["generic_by_family"].get(str(k % 7), 0) + 1
            print(f"payload {k}: no specialised launch (jit_fallbacks {st.jit_fallbacks}, handed back {st.flow_fallback}, retried {st.flow_retried})", file=sys.stderr)
        t["not_on_flow_kernel"] += int(st.flow_scenarios == 0)
        assert int(gen.engine_stats.specialised_launches) == 0
        try:
            assert np.array_equal(res.counts[:, :6], gen.counts[:, :6]), (k, "counts")
            assert np.array_equal(res.counts[:, _abi.CNT_MARKS], gen.counts[:, _abi.CNT_MARKS]), (k, "marks")
            for i in range(N):
                assert np.array_equal(res[i].rqs_clock.view(np.uint64), gen[i].rqs_clock.view(np.uint64)), (k, i, "rqs_clock")
                assert np.array_equal(res[i]._samples, gen[i]._samples), (k, i, "samples")  # noqa: SLF001
            plan = lower(make(k))
            for i in (0, N - 1):
                want = ol.simulate(plan, int(res.seeds[i]))
                assert np.array_equal(res[i].counts[:5].astype(np.uint64), want.counts[:5]), (k, i, "oracle counts")
                assert np.array_equal(res[i].rqs_clock.view(np.uint64), want.clock.view(np.uint64)), (k, i, "oracle rqs_clock")
                assert np.array_equal(res[i]._samples, want.samples), (k, i, "oracle samples")  # noqa: SLF001
                t["oracle_checks"] += 1
        except AssertionError as exc:
            failures.append(str(exc)[:300])
            print(f"DIFFERENT payload {k}: {str(exc)[:300]}", file=sys.stderr)
    t["different"] = len(failures)
    t["failures"] = failures[:10]
    print(json.dumps(t))


if __name__ == "__main__":   # (the pool's workers import this module: they must not start pools of their own)
    main()
