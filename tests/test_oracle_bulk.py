"""oracle/bulk.py (TEST INFRASTRUCTURE): the C oracle over many scenarios on every host core, digests back -- the checker of
the GPU suite's bulk parity tests must itself agree with the one-scenario oracle call it replaces."""

from __future__ import annotations

import numpy as np

from asyncflow_amd.plan import lower
from oracle import bulk
from oracle import oracle_lib as ol
from oracle.scenarios import lb_two_servers, lb_with_events


def test_pool_digests_equal_the_serial_oracle():
    payload = lb_two_servers(horizon=12)
    seeds = list(range(500, 540))
    rows = bulk.simulate_many(payload, seeds, procs=4)
    assert len(rows) == len(seeds)
    plan = lower(payload)
    for s, (counts, d_clock, d_samples, waits) in zip(seeds, rows):
        r = ol.simulate(plan, s)
        assert counts == r.counts.tolist() and waits == 0
        assert d_clock == bulk.digest_clock(r.clock) and d_samples == bulk.digest_samples(r.samples)
    assert len({row[1] for row in rows}) == len(seeds)          # (a digest that did not depend on the data would pass the above too)


def test_overrides_reach_the_workers_and_one_bit_changes_a_digest():
    payload = lb_with_events(users=60, horizon=30, scale=0.05)
    over = [[("gen_users_mean", 0, 40.0 + 10.0 * i), ("edge_mean", 2, 0.002 * (1 + i))] for i in range(6)]
    rows = bulk.simulate_many(payload, [7] * 6, over, procs=3)
    for i, (counts, d_clock, d_samples, _) in enumerate(rows):
        plan = lower(payload)
        ol.apply_overrides(plan, {(n, k): v for n, k, v in over[i]})
        r = ol.simulate(plan, 7)
        assert counts == r.counts.tolist() and d_clock == bulk.digest_clock(r.clock) and d_samples == bulk.digest_samples(r.samples)
        flipped = r.clock.copy()
        flipped.view(np.uint64)[len(flipped) // 2, 1] ^= 1
        assert bulk.digest_clock(flipped) != d_clock
    assert rows[0][0][0] < rows[5][0][0]                         # more users, more arrivals
