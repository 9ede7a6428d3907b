"""CPU differential test of the one-lane-per-scenario arrival sampler (asyncflow_amd/csrc/af_pregen.hpp).

The chain wave of `af_arrival_groups` walks a scenario's draws eight per step with the sums of the sequential sampler and a division whose
divisor half is hoisted out of the window.  tests/hostcheck compiles the very same per-lane functions with g++ and drives one
lane the way the kernel does; the arrival times must be those of `af::gen_next_gap` -- the sequential statement of
samplers/poisson_poisson.py:51-82 / gaussian_poisson.py:63-94 the oracle follows -- bit for bit.
"""

from __future__ import annotations

import numpy as np
import pytest

from asyncflow_amd import _abi
from tests.hostcheck import build as hc

POISSON, NORMAL = 0, 1


def _same(seed, **kw):
    n0, want, f0 = hc.arrivals(0, seed, **kw)
    for which in (1, 2):
        n, got, f = hc.arrivals(which, seed, **kw)
        assert (n, f) == (n0, f0), (which, seed, kw, n, n0, f, f0)
        assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), (which, seed, kw)
    return n0, f0


@pytest.mark.parametrize("dist", [POISSON, NORMAL])
def test_lane_sampler_reproduces_the_sequential_sampler(dist):
    rng = np.random.default_rng(20260922 + dist)
    total = 0
    for case in range(60):
        horizon = float(rng.choice([5, 30, 61, 120]))
        window = float(rng.choice([1, 2, 7, 60]))
        mean = float(rng.choice([0.3, 3, 40, 120]))       # 0.3 users: most windows are empty
        rpm = float(rng.choice([5, 20, 90]))
        expected = mean * rpm / 60.0 * horizon
        n_draw = int(expected * 1.5 + 6.0 * np.sqrt(expected + 1.0) + 64)
        n, flags = _same(int(rng.integers(1, 2**62)) + case, dist=dist, mean=mean, sigma=mean * 0.4, rpm=rpm,
                         window_s=window, horizon=horizon, n_draw=n_draw)
        assert flags == 0
        total += n
    assert total > 50_000


def test_lane_sampler_edge_cases():
    # the arrival array fills up: same prefix, same flag
    n, flags = _same(7, dist=POISSON, mean=50.0, sigma=0.0, rpm=60.0, window_s=5.0, horizon=30.0, n_draw=200)
    assert n == 200 and flags == _abi.FLAG_DRAW_OVERFLOW
    # exactly full without overflow is not flagged by either
    n_exact, _ = _same(7, dist=POISSON, mean=50.0, sigma=0.0, rpm=60.0, window_s=5.0, horizon=30.0, n_draw=4000)
    n, flags = _same(7, dist=POISSON, mean=50.0, sigma=0.0, rpm=60.0, window_s=5.0, horizon=30.0, n_draw=n_exact)
    assert n == n_exact
    # nobody ever active; a window longer than the horizon; a rate outside the hoisted division's range
    assert _same(3, dist=POISSON, mean=0.0, sigma=0.0, rpm=60.0, window_s=1.0, horizon=20.0, n_draw=64) == (0, 0)
    _same(11, dist=NORMAL, mean=30.0, sigma=10.0, rpm=30.0, window_s=500.0, horizon=20.0, n_draw=1024)
    _same(13, dist=POISSON, mean=3.0, sigma=0.0, rpm=1e-19 * 60.0, window_s=10.0, horizon=50.0, n_draw=64)


def test_lane_sampler_under_quantised_gaps():
    """The tie generator of the host builds (gaps rounded to 2^-bits) makes window ends and the horizon coincide with
    arrival times: `>=` against the window end and `>` against the horizon are both exercised on equality."""
    hc.set_test_quantum(4)
    try:
        for seed in range(20):
            _same(1000 + seed, dist=POISSON, mean=30.0, sigma=0.0, rpm=60.0, window_s=1.0, horizon=16.0, n_draw=1024)
    finally:
        hc.set_test_quantum(0)
