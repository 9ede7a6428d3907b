"""The analyzer restatement (oracle/analyzer_oracle.py) against what the reference's own
ResultsAnalyzer produced for every golden fixture (values stored by oracle/make_golden.py)."""

from __future__ import annotations

import json

import numpy as np
import pytest

from oracle import analyzer_oracle as ao
from tests.conftest import GOLDEN_DIR, golden_names


@pytest.mark.parametrize("name", golden_names())
def test_latency_stats_and_rps_equal_the_reference_analyzer(name):
    z = np.load(GOLDEN_DIR / f"{name}.npz", allow_pickle=False)
    stats = ao.latency_stats(z["clock"])
    assert np.array_equal(stats.view(np.uint64), z["latency_stats"].view(np.uint64)), (stats, z["latency_stats"])
    total_time = json.loads(str(z["payload_json"]))["sim_settings"]["total_simulation_time"]
    ts, rps = ao.throughput_series(z["clock"], total_time)
    assert np.array_equal(rps, z["rps"])
    assert np.array_equal(ts, np.arange(1, len(rps) + 1, dtype=np.float64))


def test_empty_and_single_completion():
    e = ao.latency_stats(np.zeros((0, 2)))
    assert e[0] == 0.0 and np.isnan(e[1:]).all()
    one = ao.latency_stats(np.array([[1.0, 1.25]]))
    assert one.tolist() == [1.0, 0.25, 0.25, 0.0, 0.25, 0.25, 0.25, 0.25]
    _, rps = ao.throughput_series(np.array([[0.0, 0.0], [0.5, 1.0], [0.5, 1.0000001], [2.0, 3.5]]), 3.0)
    assert rps.tolist() == [2.0, 1.0, 0.0]          # (k-1, k] windows, finish 3.5 is beyond the horizon


def test_histogram_and_series_spec():
    clock = np.array([[0.0, 0.0], [0.0, 0.0099], [0.0, 0.01], [0.0, 0.5], [0.0, 7.0]])
    assert ao.latency_histogram(clock, 4, 0.04).tolist() == [2, 1, 0, 2]
    mean, mx = ao.series_mean_max(np.array([[1, 2, 3, 6], [0, 0, 0, 0]], dtype=np.uint32))
    assert mean.tolist() == [3.0, 0.0] and mx.tolist() == [6, 0]
