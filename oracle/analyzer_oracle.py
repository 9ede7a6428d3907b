"""CPU restatement of the reference's post-run analysis for one scenario (TEST INFRASTRUCTURE).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product path (asyncflow_amd) never does.

Follows ``ResultsAnalyzer._process_event_metrics``
(/root/reference/src/asyncflow/metrics/analyzer.py:83-126) call for call: the same numpy
functions on the same float64 array, so the values are the reference's own.  Pinned by
tests/test_analyzer_oracle.py against the ``latency_stats`` / ``rps`` the reference's unmodified
ResultsAnalyzer produced for every golden fixture (oracle/make_golden.py).

The last two helpers are the SPEC of outputs the reference does not have (latency histogram,
per-series mean/max); they exist so that the HIP analyzer has something exact to be compared with.
"""

from __future__ import annotations

import numpy as np

LATENCY_KEYS = ("total_requests", "mean", "median", "std_dev", "p95", "p99", "min", "max")


def latency_stats(clock: np.ndarray) -> np.ndarray:
    """analyzer.py:86-106 -- [8] float64 in LatencyKey order; no completions -> total 0, rest NaN."""
    clock = np.asarray(clock, dtype=np.float64).reshape(-1, 2)
    if clock.shape[0] == 0:
        return np.array([0.0] + [np.nan] * 7)
    arr = np.array([float(f) - float(s) for s, f in clock], dtype=float)   # analyzer.py:86-89
    return np.array([
        float(arr.size),
        float(np.mean(arr)),
        float(np.median(arr)),
        float(np.std(arr)),
        float(np.percentile(arr, 95)),
        float(np.percentile(arr, 99)),
        float(np.min(arr)),
        float(np.max(arr)),
    ])


def throughput_series(clock: np.ndarray, total_time: float, window: float = 1.0) -> tuple[np.ndarray, np.ndarray]:
    """analyzer.py:108-126 -- the two-pointer walk over sorted completion times, windows (k-1, k]."""
    clock = np.asarray(clock, dtype=np.float64).reshape(-1, 2)
    completion_times = sorted(float(f) for f in clock[:, 1])
    timestamps: list[float] = []
    rps: list[float] = []
    idx = 0
    current_end = window
    while current_end <= total_time:
        count = 0
        while idx < len(completion_times) and completion_times[idx] <= current_end:
            count += 1
            idx += 1
        timestamps.append(current_end)
        rps.append(count / window)
        current_end += window
    return np.array(timestamps), np.array(rps)


def latency_histogram(clock: np.ndarray, bins: int, hist_max: float) -> np.ndarray:
    """Linear bins over [0, hist_max); bin = floor(lat * (bins / hist_max)); the last bin also
    takes everything beyond (spec of af_summary_t.hist, include/asyncflow_hip.h)."""
    clock = np.asarray(clock, dtype=np.float64).reshape(-1, 2)
    lat = clock[:, 1] - clock[:, 0]
    scale = np.float64(bins) / np.float64(hist_max)
    b = np.minimum(np.floor(lat * scale), bins - 1).astype(np.int64)
    return np.bincount(b, minlength=bins).astype(np.uint32)


def series_mean_max(samples: np.ndarray, n_edges: int | None = None) -> tuple[np.ndarray, np.ndarray]:
    """samples [n_series][ticks] 4-byte words -> mean (f64) and maximum per series.

    Rows are int32 counts, except the ram_in_use row of every server (row n_edges + 3 s + 2,
    include/asyncflow_hip.h), which holds float32 values: its mean is the f64 sum of the values /
    ticks (np.mean of the list the reference keeps, analyzer.py:127-142), its maximum the float
    maximum returned as float32 bits.  ``n_edges=None`` treats every row as integers.
    """
    samples = np.asarray(samples).view(np.uint32)
    n_series, ticks = samples.shape
    is_f = np.zeros(n_series, dtype=bool)
    if n_edges is not None:
        j = np.arange(n_series)
        is_f = (j >= n_edges) & ((j - n_edges) % 3 == 2)
    if not ticks:
        return np.full(n_series, np.nan), np.zeros(n_series, dtype=np.uint32)
    mean = samples.astype(np.uint64).sum(axis=1).astype(np.float64) / np.float64(ticks)
    mx = samples.max(axis=1).astype(np.uint32)
    for r in np.nonzero(is_f)[0]:
        vals = samples[r].view(np.float32).astype(np.float64)
        mean[r] = vals.sum() / np.float64(ticks)
        mx[r] = np.float32(vals.max()).view(np.uint32)
    return mean, mx
