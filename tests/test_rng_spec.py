"""The oracle's RNG / math spec: known-answer vectors and accuracy bounds (CPU)."""

from __future__ import annotations

import ctypes as C
import math

import numpy as np
import pytest

from oracle import oracle_lib as ol

# Random123 kat_vectors, philox4x32 10 rounds (Salmon et al., SC'11 reference data)
KAT = [
    ((0, 0, 0, 0), (0, 0), "6627e8d5 e169c58d bc57ac4c 9b00dbd8"),
    ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2, "408f276d 41c83b0e a20bc7c6 6d5451fd"),
    ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0), "d16cfe09 94fdcceb 5001e420 24126ea1"),
]


@pytest.mark.parametrize(("ctr", "key", "want"), KAT)
def test_philox_known_answers(ctr, key, want):
    out = (C.c_uint32 * 4)()
    ol.lib().orc_x_philox(*ctr, *key, out)
    assert " ".join(f"{x:08x}" for x in out) == want


def test_uniform_is_53_bit_in_unit_interval():
    L = ol.lib()
    u = np.array([L.orc_x_uniform(0x5EED0000, 3, i, j) for i in range(2000) for j in range(4)])
    assert u.min() >= 0.0 and u.max() < 1.0
    assert np.all(u * 2.0**53 == np.floor(u * 2.0**53))
    assert abs(u.mean() - 0.5) < 0.01 and abs(u.var() - 1 / 12) < 0.005
    # distinct streams / indices / seeds decorrelate
    assert L.orc_x_uniform(1, 1, 0, 0) != L.orc_x_uniform(1, 2, 0, 0) != L.orc_x_uniform(2, 1, 0, 0)


def _ulps(a: float, b: float) -> float:
    return abs(a - b) / np.spacing(abs(b)) if b != 0 else abs(a)


def test_log_exp_within_one_ulp_of_libm():
    L = ol.lib()
    rng = np.random.default_rng(1)
    xs = np.concatenate([rng.random(20000), 1 - rng.random(500) * 1e-12, np.exp(rng.uniform(-700, 700, 5000))])
    assert max(_ulps(L.orc_x_log(float(x)), math.log(x)) for x in xs) <= 1.0
    es = np.concatenate([rng.uniform(-20, 20, 20000), rng.uniform(-700, 700, 5000), rng.uniform(-1e-3, 1e-3, 500)])
    assert max(_ulps(L.orc_x_exp(float(x)), math.exp(x)) for x in es) <= 1.0
    assert L.orc_x_log(1.0) == 0.0 and L.orc_x_exp(0.0) == 1.0
    assert L.orc_x_log(0.0) == -math.inf and math.isnan(L.orc_x_log(-1.0))


def test_norminv_matches_scipy():
    from scipy.special import ndtri

    L = ol.lib()
    rng = np.random.default_rng(2)
    ps = np.concatenate([rng.random(20000), np.exp(rng.uniform(np.log(1e-300), np.log(1e-3), 5000))])
    ps = ps[(ps > 0) & (ps < 1)]
    got = np.array([L.orc_x_norminv(float(p)) for p in ps])
    want = ndtri(ps)
    assert np.max(np.abs(got - want) / np.maximum(np.abs(want), 1e-9)) < 5e-15
    assert L.orc_x_norminv(0.5) == 0.0


def test_engine_header_math_equals_the_oracle_bit_for_bit():
    """af_math.hpp (host build of the header the kernels compile) against oracle_rng.h, bit patterns: normal quantile (the
    boundaries of AS 241's three regions included), logarithm, exponential."""
    from tests.hostcheck import build as hc

    L = ol.lib()
    rng = np.random.default_rng(5)
    edge = np.array([0.075, 0.925, np.nextafter(0.075, 0), np.nextafter(0.075, 1), np.nextafter(0.925, 0), np.nextafter(0.925, 1),
                     0.5, np.nextafter(0.5, 0), np.nextafter(0.5, 1), 2.0 ** -53, 1 - 2.0 ** -53, 1.3887943864964021e-11,
                     1.4e-11, 1.38e-11, 1 - 1.4e-11, 1e-300, 5e-324, 0.0, 1.0])
    ps = np.concatenate([rng.random(200000), rng.random(20000) * 0.075, 1 - rng.random(20000) * 0.075,
                         np.exp(rng.uniform(np.log(1e-300), np.log(1e-3), 20000)), (rng.integers(0, 2 ** 53, 20000) / 2.0 ** 53), edge])
    want = np.array([L.orc_x_norminv(float(p)) for p in ps])
    assert np.array_equal(hc.math(3, ps).view(np.uint64), want.view(np.uint64))
    xs = np.concatenate([rng.random(50000), np.exp(rng.uniform(-700, 700, 5000))])
    assert np.array_equal(hc.math(1, xs).view(np.uint64), np.array([L.orc_x_log(float(x)) for x in xs]).view(np.uint64))
    es = np.concatenate([rng.uniform(-20, 20, 50000), rng.uniform(-700, 700, 5000)])
    assert np.array_equal(hc.math(2, es).view(np.uint64), np.array([L.orc_x_exp(float(x)) for x in es]).view(np.uint64))


@pytest.mark.parametrize("mean", [0.003, 0.7, 4.0, 16.0, 40.0, 400.0, 1000.0])
def test_poisson_moments(mean):
    L = ol.lib()
    n = 4000
    x = np.array([L.orc_x_poisson(mean, 77, 0, i, 0) for i in range(n)], dtype=np.float64)
    se = math.sqrt(mean / n)
    assert abs(x.mean() - mean) < 5 * se + 1e-12
    assert abs(x.var() - mean) < 0.15 * mean + 5 * se


def test_variates_follow_reference_semantics():
    """common_helpers.py:49-89: sigma=variance, lognormal mean is mu of log, uniform ignores mean."""
    L = ol.lib()
    n = 4000
    ex = np.array([L.orc_x_variate(3, 0.003, 0.0, 5, 1, i, 1) for i in range(n)])
    assert abs(ex.mean() - 0.003) < 5 * 0.003 / math.sqrt(n) and ex.min() >= 0
    nm = np.array([L.orc_x_variate(1, 10.0, 2.0, 5, 1, i, 1) for i in range(n)])
    assert abs(nm.mean() - 10.0) < 0.2 and abs(nm.std() - 2.0) < 0.1
    tr = np.array([L.orc_x_variate(1, 0.0, 1.0, 5, 1, i, 1) for i in range(n)])
    assert tr.min() == 0.0 and 0.4 < (tr == 0).mean() < 0.6  # max(0, N(0,1))
    ln = np.array([L.orc_x_variate(2, 0.001, 0.25, 5, 1, i, 1) for i in range(n)])
    assert abs(np.median(ln) - math.exp(0.001)) < 0.03  # ~1 s median (SURVEY trap 4)
    un = np.array([L.orc_x_variate(4, 123.0, 0.0, 5, 1, i, 1) for i in range(n)])
    assert 0 <= un.min() and un.max() < 1 and abs(un.mean() - 0.5) < 0.03
