"""GPU fuzz of round 6's additions to the next-event kernels: fractional RAM needs (simpy's waiting `Container.put`, dead-locked
RAM containers -- af_core.hpp::m_srv_finish / m_put_trigger) and saturated servers whose wait queues outgrow the 16 384 waiters
of rounds 1-5.  Every scenario of every payload against the ORACLE (oracle/bulk.py: every host core), which
tests/test_reference_live.py holds to the live reference on this family.

    python scripts/gpu_fuzz_frac_ram.py [payloads, default 100] [first payload index, default 0]

Prints one JSON line of tallies; `different` counts scenarios with any difference (the first ten in `failures`)."""
import json
import random
import sys
import warnings
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from asyncflow_amd import _abi  # noqa: E402
from asyncflow_amd.runner import SimulationRunner  # noqa: E402
from oracle import bulk  # noqa: E402
from oracle.scenarios import fractional_ram_fuzz  # noqa: E402


def run(n_payloads: int = 100, k0: int = 0, seeds_per_payload: int = 8) -> dict:
    t = {"payloads": 0, "scenarios": 0, "with_waiting_puts": 0, "ram_dead_or_starved": 0, "on_flow_kernel": 0, "overflow_raised": 0,
         "max_fifo_capacity": 0, "different": 0, "failures": []}
    for k in range(k0, k0 + n_payloads):
        payload = fractional_ram_fuzz(random.Random(424_200 + k), horizon=10)
        seeds = np.arange(seeds_per_payload, dtype=np.uint64) + 900 + 1000 * k
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore", RuntimeWarning)
                res = SimulationRunner(simulation_input=payload, seeds=seeds, on_negative_delay="flag").run()
        except OverflowError as exc:
            t["overflow_raised"] += 1
            print(f"payload {k}: OverflowError: {str(exc)[:160]}", file=sys.stderr)
            continue
        t["payloads"] += 1
        t["on_flow_kernel"] += int(res.engine_stats.flow_scenarios)
        t["max_fifo_capacity"] = max(t["max_fifo_capacity"], int(res.engine_stats.fifo_capacity))
        want = bulk.simulate_many(payload, [int(s) for s in seeds])
        for i in range(len(seeds)):
            w_counts, w_clock, w_samples, w_waits = want[i]
            got = res[i]
            t["scenarios"] += 1
            t["with_waiting_puts"] += w_waits > 0
            t["ram_dead_or_starved"] += (w_counts[_abi.CNT_FLAGS] & _abi.FLAG_RAM_STARVED) != 0
            ok = (got.counts[:5].astype(np.uint64).tolist() == w_counts[:5]
                  and (int(got.counts[_abi.CNT_FLAGS]) & 0xFF) == (w_counts[_abi.CNT_FLAGS] & 0xFF)
                  and bulk.digest_clock(got.rqs_clock) == w_clock and bulk.digest_samples(got._samples) == w_samples)  # noqa: SLF001
            if not ok:
                t["different"] += 1
                t["failures"].append([k, i, got.counts.tolist(), w_counts])
    t["failures"] = t["failures"][:10]
    return t


if __name__ == "__main__":
    print(json.dumps(run(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 0)))
