"""Differential pin of the SimPy stand-in (oracle/simpy_standin) against a REAL SimPy wheel, wherever one exists.

SURVEY 8c: SimPy 4.1.1 owns the event ordering of the reference and is not vendored; the build container and the GPU box
have no wheel, so the oracle's chain rests on a restatement of SimPy's published scheduling rules that is pinned by the
reference's own 183 tests.  This file closes the remaining "unpinned" wherever `import simpy` yields a genuine 4.x: random
process graphs over exactly the primitives the reference uses -- Timeout, Process (Initialize), Event.succeed, Store
put / get, Container put / get, run(until=...) -- with delays drawn from a tiny set of dyadic values so that EXACT
timestamp ties happen all the time, executed on both kernels; the traces (time, process, step, value) and the order in
which `step()` pops events must be identical.  Self-contained: no AsyncFlow sources are needed.

Without a real wheel the comparison is skipped WITH THE REASON RECORDED; what still runs everywhere: the programs are
deterministic on the stand-in, and a handful of orderings SimPy documents are asserted on it directly.
"""

from __future__ import annotations

import importlib
import importlib.util
import random
import sys
from pathlib import Path

import pytest

STANDIN_DIR = Path(__file__).resolve().parent.parent / "oracle" / "simpy_standin"


def _load_standin():
    spec = importlib.util.spec_from_file_location("af_simpy_standin", STANDIN_DIR / "simpy" / "__init__.py")
    mod = importlib.util.module_from_spec(spec)
    sys.modules["af_simpy_standin"] = mod
    spec.loader.exec_module(mod)
    return mod


def _load_real():
    """A genuine SimPy 4.x, or (None, reason)."""
    try:
        import importlib.metadata as md

        version = md.version("simpy")
    except Exception as exc:  # noqa: BLE001 - not installed
        return None, f"no simpy distribution installed ({type(exc).__name__})"
    for name in [k for k in sys.modules if k == "simpy" or k.startswith("simpy.")]:
        if "simpy_standin" in (getattr(sys.modules[name], "__file__", "") or ""):
            del sys.modules[name]          # another test put the stand-in first on sys.path: drop it, import the wheel
    path = [p for p in sys.path if "simpy_standin" not in p]
    old = sys.path[:]
    try:
        sys.path[:] = path
        mod = importlib.import_module("simpy")
    finally:
        sys.path[:] = old
    if "standin" in getattr(mod, "__version__", "") or not version.startswith("4."):
        return None, f"simpy {version} is not a genuine 4.x wheel"
    return mod, version


DELAYS = (0.0, 0.25, 0.5, 0.5, 1.0, 1.0, 1.5)     # dyadic: sums are exact, ties by the hundred


def run_program(simpy, seed: int, n_proc: int = 7, n_ops: int = 14, until: float = 9.0):
    """One random process graph; returns (trace, pop order).  The PROGRAM is a function of `seed` only; every decision
    that depends on the run is taken from values both kernels must agree on (what a get returned, the clock)."""
    rng = random.Random(seed)
    env = simpy.Environment()
    stores = [simpy.Store(env) for _ in range(2)]
    boxes = [simpy.Container(env, capacity=10, init=rng.choice((0, 3, 10))) for _ in range(2)]
    shared = [env.event() for _ in range(3)]
    trace: list[tuple] = []
    programs = [[(rng.choice(("timeout", "timeout", "put", "get", "cget", "cput", "spawn", "wait", "fire")),
                  rng.choice(DELAYS), rng.randrange(2), rng.randrange(1, 5), rng.randrange(3)) for _ in range(n_ops)]
                for _ in range(n_proc)]

    def child(pid: int, k: int, delay: float):
        trace.append((env.now, pid, k, "child-start"))
        yield env.timeout(delay)
        trace.append((env.now, pid, k, "child-end"))
        return pid * 100 + k

    def proc(pid: int):
        for k, (op, delay, which, amount, ev) in enumerate(programs[pid]):
            if op == "timeout":
                got = yield env.timeout(delay, value=k)
            elif op == "put":
                got = yield stores[which].put((pid, k))
            elif op == "get":
                got = yield stores[which].get() | env.timeout(delay + 0.5)       # never blocks for good
                got = sorted(str(v) for v in got.values())
            elif op == "cget":
                req = boxes[which].get(amount)
                got = yield req | env.timeout(delay + 0.5)
                if req not in got:
                    req.cancel()
                got = req in got
            elif op == "cput":
                req = boxes[which].put(amount)
                got = yield req | env.timeout(delay + 0.25)
                if req not in got:
                    req.cancel()
                got = req in got
            elif op == "spawn":
                got = yield env.process(child(pid, k, delay))
            elif op == "wait":
                got = yield shared[ev] | env.timeout(delay + 1.0)
                got = shared[ev] in got
            else:  # fire
                if not shared[ev].triggered:
                    shared[ev].succeed(value=(pid, k))
                got = yield env.timeout(0)
            trace.append((env.now, pid, k, op, repr(got), boxes[0].level, boxes[1].level, len(stores[0].items), len(stores[1].items)))

    for pid in range(n_proc):
        env.process(proc(pid))
    pops: list[tuple] = []
    while env.peek() < until:
        pops.append((env.peek(), len(trace)))
        env.step()
    return trace, pops


def test_programs_are_deterministic_on_the_standin_and_tie_heavy():
    st = _load_standin()
    ties = 0
    for seed in range(40):
        a = run_program(st, seed)
        b = run_program(st, seed)
        assert a == b
        times = [p[0] for p in a[1]]
        ties += sum(1 for x, y in zip(times, times[1:]) if x == y)
        assert times == sorted(times)
    assert ties > 2000        # the point of the dyadic delays: most pops share their timestamp with a neighbour


def test_documented_orderings_hold_on_the_standin():
    """What SimPy's documentation and source state outright: heap key (time, priority, insertion id);
    a Process starts through an URGENT Initialize at `now`; Timeout(0) is NORMAL at `now`; Store is FIFO; Container gets
    are FIFO with head-of-line blocking; run(until=t) stops BEFORE events at t."""
    simpy = _load_standin()
    env = simpy.Environment()
    log = []

    def a():
        log.append("a0")
        yield env.timeout(0)
        log.append("a1")

    def b():
        log.append("b0")
        yield env.timeout(0)
        log.append("b1")

    env.process(a())
    env.timeout(0).callbacks.append(lambda _e: log.append("t"))     # created after a's Initialize, before b's
    env.process(b())
    env.run(until=1)
    assert log == ["a0", "b0", "t", "a1", "b1"]       # both Initialize events (URGENT) before the NORMAL timeout
    env = simpy.Environment()
    box = simpy.Container(env, capacity=10, init=2)
    got = []

    def taker(name, amount):
        yield box.get(amount)
        got.append((name, env.now))

    def giver():
        yield env.timeout(1)
        yield box.put(1)          # level 3: the head waiter (5) still blocks the small one behind it
        yield env.timeout(1)
        yield box.put(4)

    env.process(taker("big", 5))
    env.process(taker("small", 1))
    env.process(giver())
    env.run(until=5)
    assert got == [("big", 2), ("small", 2)]          # head-of-line blocking, then both at the same instant, FIFO
    env = simpy.Environment()
    fired = []
    env.timeout(3).callbacks.append(lambda _e: fired.append(env.now))
    env.run(until=3)
    assert fired == [] and env.now == 3               # the stop event is URGENT at t: events AT t do not run


def test_standin_equals_a_real_simpy_wheel_on_random_process_graphs():
    real, why = _load_real()
    if real is None:
        pytest.skip(f"real-SimPy pin not possible here: {why} (the stand-in stays pinned by the reference's own 183 tests)")
    st = _load_standin()
    for seed in range(300):
        assert run_program(st, seed) == run_program(real, seed), f"program {seed}: the stand-in and simpy {why} disagree"
