#!/usr/bin/env python3
"""Throughput of SEVERAL sweeps in flight on one GPU (round 6, a measurement beside the bench line, never the bench line).

BASELINE config 2 is 2.44 residency rounds of the stage-parallel kernel: the last 0.44 round leaves most of the chip idle
(the analyzer already runs beside it).  Inside ONE sweep that tail cannot be filled -- every remaining scenario is
resident -- but a user who runs sweep after sweep (a study of several payloads) can keep `k` engines, each with its own HIP
streams and output buffers, and call them from `k` host threads (ctypes releases the GIL inside `af_engine_run_summarized`): the
next sweep's arrival pre-generation and first waves then run beside the previous sweep's tail.

    python scripts/gpu_two_sweeps_in_flight.py [--config 2] [--steps 10] [--in-flight 2]

prints one JSON object: ms per sweep with 1 and with `k` sweeps in flight (same total number of sweeps), the results of the
concurrent sweeps compared bit for bit with the lone run's (stats, histogram, counts).
"""
from __future__ import annotations

import argparse
import json
import sys
import threading
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import bench  # noqa: E402


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--steps", type=int, default=10, help="sweeps per engine in the concurrent leg (the lone leg runs steps x in-flight)")
    ap.add_argument("--in-flight", type=int, default=2)
    ap.add_argument("--stagger", type=float, default=-1.0,
                    help="ms between the threads' first sweeps (-1: lone ms / in-flight, so that one sweep's last residency round "
                         "meets the other's full ones; 0: all start together and run in lock-step, tails together)")
    a = ap.parse_args()
    args = bench.make_parser().parse_args(["--config", str(a.config), "--no-cpu-baseline", "--no-diagnostics"])
    args.horizon = None

    import torch

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    wl = bench.build_workload(a.config, 0, 1, 0, None)
    sweeps = [bench.RankSweep(wl, dev, args) for _ in range(a.in_flight)]
    for sw in sweeps:
        sw.prepare()
        sw.step()
    torch.cuda.synchronize(dev)

    total = a.steps * a.in_flight
    t0 = time.perf_counter()
    for _ in range(total):
        sweeps[0].step()
    torch.cuda.synchronize(dev)
    lone_ms = (time.perf_counter() - t0) * 1e3 / total
    want = [t.clone() for t in (sweeps[0].s_stats, sweeps[0].s_hist, sweeps[0].counts)]

    errors: list[BaseException] = []

    stagger_ms = a.stagger if a.stagger >= 0.0 else lone_ms / a.in_flight

    def work(sw: "bench.RankSweep", slot: int) -> None:
        try:
            time.sleep(slot * stagger_ms * 1e-3)
            for _ in range(a.steps):
                sw.step()
        except BaseException as exc:  # noqa: BLE001
            errors.append(exc)

    threads = [threading.Thread(target=work, args=(sw, k)) for k, sw in enumerate(sweeps)]
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize(dev)
    many_ms = (time.perf_counter() - t0) * 1e3 / total
    if errors:
        raise errors[0]
    same = all(torch.equal(w.view(torch.int64) if w.dtype == torch.float64 else w, (g.view(torch.int64) if g.dtype == torch.float64 else g))
               for sw in sweeps for w, g in zip(want, (sw.s_stats, sw.s_hist, sw.counts)))
    print(json.dumps({"config": a.config, "scenarios_per_sweep": sweeps[0].n, "sweeps_timed": total, "in_flight": a.in_flight,
                      "ms_per_sweep_one_in_flight": lone_ms, f"ms_per_sweep_{a.in_flight}_in_flight": many_ms,
                      "stagger_ms": stagger_ms, "gain": lone_ms / many_ms, "results_identical_to_the_lone_run": bool(same)}))
    return 0 if same else 1


if __name__ == "__main__":
    raise SystemExit(main())
