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
