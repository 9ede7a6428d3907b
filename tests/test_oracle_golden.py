"""Pin the C oracle against the committed golden vectors (CPU, bit-exact).

The fixtures under tests/golden/ were produced by oracle/make_golden.py from the
UNMODIFIED reference actors (imported from /root/reference in the build
container).  They are the reference's behaviour; the C restatement must
reproduce every count, every (start, finish) pair and every sample bit for bit.
"""

from __future__ import annotations

import json

import numpy as np
import pytest

from asyncflow_amd.plan import lower
from oracle import oracle_lib as ol
from tests.conftest import GOLDEN_DIR, golden_names


def load_fixture(name: str) -> dict:
    z = np.load(GOLDEN_DIR / f"{name}.npz", allow_pickle=False)
    return {k: z[k] for k in z.files}


def test_fixtures_exist():
    assert len(golden_names()) >= 7


@pytest.mark.parametrize("name", golden_names())
def test_oracle_reproduces_reference_bit_for_bit(name):
    fx = load_fixture(name)
    payload = json.loads(str(fx["payload_json"]))
    plan = lower(payload)
    res = ol.simulate(plan, int(fx["seed"]))
    assert res.generated == int(fx["generated"])
    assert res.completed == int(fx["completed"])
    assert res.dropped == int(fx["dropped"])
    assert res.ticks == int(fx["ticks"]) == plan.tick_count
    assert res.clock.shape == fx["clock"].shape
    assert np.array_equal(res.clock.view(np.uint64), fx["clock"].view(np.uint64)), "rqs_clock differs"
    assert np.array_equal(res.samples, fx["samples"]), "sampled series differ"
    assert plan.edge_ids == json.loads(str(fx["edge_ids"]))
    assert plan.server_ids == json.loads(str(fx["server_ids"]))


@pytest.mark.parametrize("name", golden_names())
def test_glibc_log_only_moves_last_bits(name):
    """Substituting the spec's log for glibc's in the arrival sampler (both < 1 ulp)
    changes no count and moves no timestamp by more than 1e-9 s."""
    fx = load_fixture(name)
    want = [int(fx["generated"]), int(fx["completed"]), int(fx["dropped"]), int(fx["ticks"])]
    assert list(fx["glibc_log_counts"]) == want
    assert float(fx["glibc_log_max_abs_delta"]) < 1e-9


def test_tick_count_follows_repeated_addition():
    """period 0.05 & T=600 -> 11 999 ticks; 0.01 & T=50 -> 5 000."""
    L = ol.lib()
    assert L.orc_tick_count(0.05, 600.0) == 11999
    assert L.orc_tick_count(0.05, 500.0) == 9999
    assert L.orc_tick_count(0.01, 50.0) == 5000


def test_oracle_is_deterministic_and_seed_sensitive():
    from oracle.scenarios import lb_two_servers

    plan = lower(lb_two_servers(horizon=10))
    a, b, c = ol.simulate(plan, 1), ol.simulate(plan, 1), ol.simulate(plan, 2)
    assert np.array_equal(a.clock, b.clock) and np.array_equal(a.samples, b.samples)
    assert not np.array_equal(a.clock[:50], c.clock[:50])


@pytest.mark.parametrize("name", [n for n in golden_names() if n.startswith("frac_ram")])
def test_ram_in_use_is_the_references_f64_value_rounded_once_to_f32(name):
    """`ram_in_use` is stored as float32 words while the reference's collector appends the server's
    `int | float` (server.py:65; `necessary_ram` may be a float, schemas/topology/endpoint.py:26).  The stated tolerance:
    every emitted word is the reference's f64 value ROUNDED ONCE -- relative error <= 2^-24, exact for whole megabytes --
    never an f32 accumulation.  The fixtures of the fractional-RAM plans hold the reference's own f64 series (`ram_f64`);
    the oracle, and through tests/test_gpu_parity.py the device, reproduce the words bit for bit."""
    fx = load_fixture(name)
    payload = json.loads(str(fx["payload_json"]))
    plan = lower(payload)
    ram_rows = [plan.n_edges + 3 * s + 2 for s in range(len(plan.server_ids))]
    f64 = fx["ram_f64"]
    assert f64.shape == (len(ram_rows), int(fx["ticks"])) and np.any(f64 != np.floor(f64))
    words = fx["samples"][ram_rows]
    assert np.array_equal(f64.astype(np.float32).view(np.uint32), words)                 # rounded once, nothing else
    decoded = words.view(np.float32).astype(np.float64)
    assert np.all(np.abs(decoded - f64) <= 2.0 ** -24 * np.abs(f64))
    res = ol.simulate(plan, int(fx["seed"]))
    assert np.array_equal(res.samples[ram_rows], words)
    if "dyadic" in name:      # multiples of 1/256 MB below 2^16 MB are f32 numbers: nothing is lost at all
        assert np.array_equal(decoded, f64)


def test_a_waiting_ram_put_is_reproduced():
    """simpy's `Container._do_put` succeeds only `if self._capacity - self._level >= event.amount`; for a fractional need that is
    false by ONE ROUNDING -- 2048 - fl(2048 - 100.3) < 100.3 -- so in the reference the request that gives 100.3 MB back waits
    until the next RAM get of that server is processed, and its response leaves that much later (server.py:270-276).  Round 5
    reported such scenarios (AF_FLAG_RAM_PUT_BLOCKED); since round 6 oracle and engine model the put queue
    (des_oracle.c::ram_trigger_put, af_core.hpp::m_put_trigger) and the reference's outputs are ordinary parity fixtures:
    frac_ram_waiting_put_t20 (586 waiting puts) and ram_put_deadlock_t20 (a waiting put facing a waiter that does not fit:
    the server's RAM is dead for the rest of the run, AF_FLAG_RAM_STARVED like the starved case of ram_starved_t30)."""
    from asyncflow_amd import _abi

    assert 2048.0 - (2048.0 - 100.3) < 100.3 and not 2048.0 - (2048.0 - 64.7) < 64.7          # the rounding itself
    assert not hasattr(_abi, "FLAG_RAM_PUT_BLOCKED")
    seen = {}
    for name in ("frac_ram_waiting_put_t20", "ram_put_deadlock_t20", "ram_starved_t30", "frac_ram_dyadic_t20"):
        fx = load_fixture(name)
        res = ol.simulate(lower(json.loads(str(fx["payload_json"]))), int(fx["seed"]))
        assert np.array_equal(res.clock, fx["clock"]) and np.array_equal(res.samples, fx["samples"])
        seen[name] = (res.put_waits, int(res.counts[_abi.CNT_FLAGS]))
    assert seen["frac_ram_waiting_put_t20"][0] > 100 and seen["frac_ram_waiting_put_t20"][1] == 0
    assert seen["ram_put_deadlock_t20"][0] >= 2 and seen["ram_put_deadlock_t20"][1] == _abi.FLAG_RAM_STARVED
    assert seen["ram_starved_t30"] == (0, _abi.FLAG_RAM_STARVED)
    assert seen["frac_ram_dyadic_t20"] == (0, 0)          # multiples of 1/256 MB: exact sums, no put ever waits
