"""Results of a batched run, with the reference analyzer's accessors per scenario.

``ScenarioResults`` restates the compute part of the reference's
``ResultsAnalyzer`` (/root/reference/src/asyncflow/metrics/analyzer.py:75-244)
on the arrays the engine wrote: same numpy calls, same bucket rule, same keys,
so a user of ``SimulationRunner(...).run()`` can keep calling
``get_latency_stats / get_throughput_series / get_sampled_metrics / get_series``.
Plotting (analyzer.py:249-589) is presentation and out of scope: the reference's
own plot helpers accept these objects through ``to_reference_analyzer``.
"""

from __future__ import annotations

from collections import defaultdict
from typing import Any, Iterator

import numpy as np

from . import _abi
from .plan import DevicePlan

LATENCY_KEYS = ("total_requests", "mean", "median", "std_dev", "p95", "p99", "min", "max")
Series = tuple[list[float], list[float]]


class ScenarioResults:
    """One scenario of a sweep; API of the reference's ``ResultsAnalyzer``."""

    _WINDOW_SIZE_S: float = 1.0

    def __init__(self, plan: DevicePlan, counts: np.ndarray, clock: np.ndarray, samples: np.ndarray | None) -> None:
        self._plan = plan
        self.counts = counts
        self.rqs_clock = clock            # float64 [completed, 2] (start, finish)
        self._samples = samples           # uint32 [n_series, ticks] raw words or None
        self.latencies: np.ndarray | None = None
        self.latency_stats: dict[str, float] | None = None
        self.throughput_series: Series | None = None
        self.sampled_metrics: dict[str, dict[str, list[float]]] | None = None

    # ---- counters -------------------------------------------------------------
    @property
    def total_generated(self) -> int:
        return int(self.counts[_abi.CNT_GENERATED])

    @property
    def total_completed(self) -> int:
        return int(self.counts[_abi.CNT_COMPLETED])

    @property
    def total_dropped(self) -> int:
        return int(self.counts[_abi.CNT_DROPPED])

    @property
    def request_events(self) -> int:
        return int(self.counts[_abi.CNT_EVENTS])

    @property
    def flags(self) -> int:
        return int(self.counts[_abi.CNT_FLAGS])

    # ---- analyzer.py:75-142 -----------------------------------------------------
    def process_all_metrics(self) -> None:
        if self.latency_stats is None and len(self.rqs_clock):
            self._process_event_metrics()
        if self.sampled_metrics is None:
            self._extract_sampled_metrics()

    def _process_event_metrics(self) -> None:
        start, finish = self.rqs_clock[:, 0], self.rqs_clock[:, 1]
        arr = finish - start
        self.latencies = arr
        if arr.size:
            self.latency_stats = {
                "total_requests": float(arr.size),
                "mean": float(np.mean(arr)),
                "median": float(np.median(arr)),
                "std_dev": float(np.std(arr)),
                "p95": float(np.percentile(arr, 95)),
                "p99": float(np.percentile(arr, 99)),
                "min": float(np.min(arr)),
                "max": float(np.max(arr)),
            }
        else:
            self.latency_stats = {}
        self.throughput_series = self._throughput(self._WINDOW_SIZE_S)

    def _throughput(self, window_s: float) -> Series:
        """Buckets (k-1, k]*window counting ``finish <= k*window`` (analyzer.py:107-125)."""
        completion = np.sort(self.rqs_clock[:, 1])
        end_time = self._plan.total_time
        timestamps: list[float] = []
        current_end = float(window_s)
        while current_end <= end_time:  # same float accumulation as the reference
            timestamps.append(current_end)
            current_end += float(window_s)
        edges = np.searchsorted(completion, np.asarray(timestamps), side="right")
        counts = np.diff(np.concatenate([[0], edges]))
        return timestamps, [float(c) / float(window_s) for c in counts]

    def _extract_sampled_metrics(self) -> None:
        metrics: dict[str, dict[str, list[float]]] = defaultdict(dict)
        plan, s = self._plan, self._samples
        enabled = set(plan.payload["sim_settings"]["enabled_sample_metrics"])
        servers_sampled = {"ready_queue_len", "event_loop_io_sleep", "ram_in_use"} <= enabled
        E = plan.n_edges
        for v, sid in enumerate(plan.server_ids):
            rows = {
                "ready_queue_len": (E + 3 * v, False),
                "event_loop_io_sleep": (E + 3 * v + 1, False),
                "ram_in_use": (E + 3 * v + 2, True),
            }
            for name, (row, is_float) in rows.items():
                if name not in enabled:
                    continue
                if s is None or not servers_sampled:
                    metrics[name][sid] = []
                elif is_float:
                    metrics[name][sid] = s[row].view(np.float32).astype(np.float64).tolist()
                else:
                    metrics[name][sid] = s[row].view(np.int32).tolist()
        if "edge_concurrent_connection" in enabled:
            for e, eid in enumerate(plan.edge_ids):
                metrics["edge_concurrent_connection"][eid] = [] if s is None else s[e].view(np.int32).tolist()
        self.sampled_metrics = metrics

    # ---- analyzer.py:147-244 (public accessors) -----------------------------------
    def list_server_ids(self) -> list[str]:
        return list(self._plan.server_ids)

    def get_latency_stats(self) -> dict[str, float]:
        self.process_all_metrics()
        return self.latency_stats or {}

    def format_latency_stats(self) -> str:
        stats = self.get_latency_stats()
        if not stats:
            return "Latency stats: (empty)"
        lines = ["════════ LATENCY STATS ════════"]
        lines.extend(f"{k.upper():<20} = {stats[k]:.6f}" for k in LATENCY_KEYS if k in stats)
        return "\n".join(lines)

    def get_throughput_series(self, window_s: float | None = None) -> Series:
        self.process_all_metrics()
        if window_s is None or window_s == self._WINDOW_SIZE_S:
            return self.throughput_series or ([], [])
        return self._throughput(float(window_s))

    def get_sampled_metrics(self) -> dict[str, dict[str, list[float]]]:
        self.process_all_metrics()
        assert self.sampled_metrics is not None
        return self.sampled_metrics

    def get_metric_map(self, key: Any) -> dict[str, list[float]]:
        self.process_all_metrics()
        assert self.sampled_metrics is not None
        name = getattr(key, "value", key)
        return self.sampled_metrics.get(name, {}) or {}

    def get_series(self, key: Any, entity_id: str) -> Series:
        vals = self.get_metric_map(key).get(entity_id, [])
        times = (np.arange(len(vals)) * self._plan.sample_period).tolist()
        return times, vals

    # ---- bridge to the reference's own analyzer / plots ----------------------------
    def to_reference_analyzer(self) -> Any:
        """Hydrate the reference's ``ResultsAnalyzer`` via duck-typed shims (needs `asyncflow`).

        Same trick as the reference's tests/unit/metrics/test_analyzer.py:34-95.
        """
        from types import SimpleNamespace

        from asyncflow.config.constants import SampledMetricName  # type: ignore[import-not-found]
        from asyncflow.metrics.analyzer import ResultsAnalyzer  # type: ignore[import-not-found]

        sm = self.get_sampled_metrics()
        client = SimpleNamespace(rqs_clock=[SimpleNamespace(start=float(a), finish=float(b)) for a, b in self.rqs_clock])
        servers = [
            SimpleNamespace(
                server_config=SimpleNamespace(id=sid),
                enabled_metrics={SampledMetricName(k): v[sid] for k, v in sm.items() if sid in v},
            )
            for sid in self._plan.server_ids
        ]
        edges = [
            SimpleNamespace(
                edge_config=SimpleNamespace(id=eid),
                enabled_metrics={SampledMetricName(k): v[eid] for k, v in sm.items() if eid in v},
            )
            for eid in self._plan.edge_ids
        ]
        settings = SimpleNamespace(
            total_simulation_time=int(self._plan.total_time), sample_period_s=self._plan.sample_period
        )
        return ResultsAnalyzer(client=client, servers=servers, edges=edges, settings=settings)


class BatchedResults:
    """All scenarios of a sweep; device-resident until a scenario is read."""

    def __init__(self, plan: DevicePlan, seeds: np.ndarray, counts: Any, clock: Any, samples: Any,
                 stats: _abi.AfStats, wall_s: float, overrides: dict[str, np.ndarray] | None = None, *,
                 online_hist: Any = None, online_rps: Any = None, online_hist_max: float = 0.0) -> None:
        self.plan = plan
        self.seeds = seeds
        self._counts_t, self._clock_t, self._samples_t = counts, clock, samples
        self.counts = counts.cpu().numpy().view(np.uint32)
        self.kernel_ms = float(stats.kernel_ms)
        self.engine_stats = stats
        self.wall_s = wall_s
        self.overrides = overrides or {}
        #: kernel-side summary (SimulationRunner(online_summary=...)): int32 [n, bins] / [n, floor(T)] on the device
        self.online_hist, self.online_rps, self.online_hist_max = online_hist, online_rps, float(online_hist_max)
        self._summ_engine: Any = None      # one af_engine_t serves every summary() call of this object
        #: '' when the stage-parallel kernel ran the plan, else why the next-event kernels did (Engine.flow_reason())
        self.flow_reason: str = ""

    def close(self) -> None:
        """Release the analyzer engine kept by :meth:`summary` (also done on garbage collection)."""
        if self._summ_engine is not None:
            self._summ_engine.close()
            self._summ_engine = None

    def __del__(self) -> None:  # pragma: no cover - best effort
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def __len__(self) -> int:
        return int(self.counts.shape[0])

    @property
    def flags(self) -> np.ndarray:
        return self.counts[:, _abi.CNT_FLAGS]

    @property
    def request_events(self) -> np.ndarray:
        return self.counts[:, _abi.CNT_EVENTS].astype(np.int64)

    def raise_on_overflow(self) -> None:
        bad = np.nonzero(self.flags & _abi.FATAL_FLAGS)[0]
        if len(bad):
            f = int(np.bitwise_or.reduce(self.flags[bad]))
            why = "; ".join(v for k, v in _abi.FLAG_NAMES.items() if f & k & _abi.FATAL_FLAGS)
            msg = f"{len(bad)} scenario(s) overflowed an engine capacity (first: #{int(bad[0])}): {why}"
            raise OverflowError(msg)

    def raise_on_negative_delay(self) -> None:
        """The reference's error behaviour for a send whose ``transit + spike`` is negative (the residue of overlapping
        spikes under a zero transit time): ``env.timeout`` raises ``ValueError("Negative delay ...")`` (edge.py:107) and
        the run produces nothing.  The engine reports such scenarios (``AF_FLAG_NEGATIVE_DELAY``); this raises for them."""
        bad = np.nonzero(self.flags & _abi.FLAG_NEGATIVE_DELAY)[0]
        if len(bad):
            msg = (f"Negative delay: {len(bad)} scenario(s) sent a message with transit + spike < 0 (first: #{int(bad[0])}, seed "
                   f"{int(self.seeds[bad[0]])}); the reference raises the same error for them "
                   "(SimulationRunner(on_negative_delay='flag') keeps the results and the flag)")
            raise ValueError(msg)

    def __getitem__(self, i: int) -> ScenarioResults:
        i = int(i)
        if not 0 <= i < len(self):
            raise IndexError(i)
        counts = self.counts[i]
        n = min(int(counts[_abi.CNT_COMPLETED]), int(self._clock_t.shape[1])) if self._clock_t is not None else 0
        clock = self._clock_t[i, :n].cpu().numpy() if self._clock_t is not None else np.zeros((0, 2))
        samples = None
        if self._samples_t is not None:  # device layout [tick][series_pitch] -> [series][tick]
            k = min(int(counts[_abi.CNT_TICKS]), int(self._samples_t.shape[1]))
            rows = self._samples_t[i, :k, : self.plan.n_series].cpu().numpy().view(np.uint32)
            samples = np.ascontiguousarray(rows.T)
        return ScenarioResults(self.plan, counts, clock, samples)

    def __iter__(self) -> Iterator[ScenarioResults]:
        return (self[i] for i in range(len(self)))

    # ---- device-side reduction of every scenario at once ---------------------------
    def summary(self, rps: bool = True, hist_bins: int = 0, hist_max: float = 0.0,
                series: bool = False) -> dict[str, Any]:
        """Per-scenario latency stats, 1-s RPS series, optional latency histogram and
        per-series mean/max, computed by the HIP analyzer (``af_engine_summarize``).

        Returns torch tensors on the run's device: ``stats`` float64 [n, 8] in
        LATENCY_KEYS order (= ``ResultsAnalyzer.get_latency_stats`` per scenario,
        metrics/analyzer.py:83-104), ``rps`` float32 [n, floor(T)] (analyzer.py:108-126),
        ``hist`` int32 [n, hist_bins], ``series_mean`` float64 / ``series_max`` int32
        [n, n_series] (the ram_in_use columns of ``series_max`` hold float32 BITS, like the sample
        words they are the maximum of: decode with :meth:`decode_series_max`).  Every statistic is bit-equal to numpy's (round 6:
        mean and std_dev too -- the kernel adds in numpy's own order, af_summary.hpp).
        A run made with ``SimulationRunner(summary=...)`` computed the summary in the engine call of the simulation itself
        (``af_engine_run_summarized``): the same arguments return those tensors.
        """
        import torch

        from .engine import Engine

        pre = getattr(self, "_summary_from_run", None)
        if pre is not None and pre["_kw"] == {"rps": rps, "hist_bins": hist_bins, "hist_max": hist_max, "series": series}:
            return {k: v for k, v in pre.items() if k != "_kw"}

        if self._clock_t is None:
            if self.online_hist is not None:
                return self._summary_from_online()
            msg = "run(collect_clock=False) kept no rqs_clock (pass online_summary=... to keep a kernel-side summary)"
            raise RuntimeError(msg)
        if series and self._samples_t is None:
            msg = "run(collect_samples=False) kept no sampled series"
            raise RuntimeError(msg)
        clock = self._clock_t
        n, cap = int(clock.shape[0]), int(clock.shape[1])
        dev = clock.device
        T = int(self.plan.total_time)
        stats = torch.empty((n, 8), dtype=torch.float64, device=dev)
        rps_t = torch.empty((n, T), dtype=torch.float32, device=dev) if rps and T > 0 else None
        hist_t = torch.empty((n, hist_bins), dtype=torch.int32, device=dev) if hist_bins else None
        smean = torch.empty((n, self.plan.n_series), dtype=torch.float64, device=dev) if series else None
        smax = torch.empty((n, self.plan.n_series), dtype=torch.int32, device=dev) if series else None
        torch.cuda.synchronize(dev)
        if self._summ_engine is None:
            self._summ_engine = Engine(self.plan, dev.index if dev.index is not None else torch.cuda.current_device())
        st = self._summ_engine.summarize(
            n,
            clock_ptr=clock.data_ptr(), clock_capacity=cap,
            samples_ptr=self._samples_t.data_ptr() if self._samples_t is not None else 0,
            tick_capacity=int(self._samples_t.shape[1]) if self._samples_t is not None else 0,
            counts_ptr=self._counts_t.data_ptr(),
            stats_ptr=stats.data_ptr(),
            rps_ptr=rps_t.data_ptr() if rps_t is not None else 0, rps_buckets=T if rps_t is not None else 0,
            hist_ptr=hist_t.data_ptr() if hist_t is not None else 0, hist_bins=hist_bins, hist_max=hist_max,
            series_mean_ptr=smean.data_ptr() if smean is not None else 0,
            series_max_ptr=smax.data_ptr() if smax is not None else 0,
        )
        out: dict[str, Any] = {"stats": stats, "keys": LATENCY_KEYS, "summary_ms": float(st.summary_ms)}
        if rps_t is not None:
            out["rps"] = rps_t
        if hist_t is not None:
            out["hist"] = hist_t
        if series:
            out["series_mean"], out["series_max"] = smean, smax
        return out

    def _summary_from_online(self) -> dict[str, Any]:
        """Latency statistics read from the kernel-side histogram: total exact, everything else accurate
        to one bin (mean / std from bin centres, percentiles by linear interpolation inside the bin,
        min / max = edges of the extreme occupied bins)."""
        import torch

        h = self.online_hist.to(torch.float64)
        n, bins = h.shape
        width = self.online_hist_max / bins
        centres = (torch.arange(bins, device=h.device, dtype=torch.float64) + 0.5) * width
        total = h.sum(dim=1)
        safe = total.clamp(min=1.0)
        mean = (h * centres).sum(dim=1) / safe
        var = (h * (centres.unsqueeze(0) - mean.unsqueeze(1)) ** 2).sum(dim=1) / safe
        cdf = h.cumsum(dim=1)

        def pct(q: float) -> Any:
            target = (q / 100.0) * total
            idx = torch.searchsorted(cdf, target.unsqueeze(1).contiguous(), right=False).squeeze(1).clamp(max=bins - 1)
            below = torch.where(idx > 0, cdf.gather(1, (idx - 1).clamp(min=0).unsqueeze(1)).squeeze(1), torch.zeros_like(total))
            inside = h.gather(1, idx.unsqueeze(1)).squeeze(1).clamp(min=1.0)
            return (idx.to(torch.float64) + ((target - below) / inside).clamp(0.0, 1.0)) * width

        occupied = h > 0
        first = torch.where(occupied.any(dim=1), occupied.to(torch.int64).argmax(dim=1), torch.zeros(n, dtype=torch.int64, device=h.device))
        last = bins - 1 - occupied.flip(dims=[1]).to(torch.int64).argmax(dim=1)
        stats = torch.stack([total, mean, pct(50.0), var.sqrt(), pct(95.0), pct(99.0), first.to(torch.float64) * width,
                             (last.to(torch.float64) + 1.0) * width], dim=1)
        stats = torch.where((total > 0).unsqueeze(1), stats, torch.full_like(stats, float("nan")))
        stats[:, 0] = total
        out: dict[str, Any] = {"stats": stats, "keys": LATENCY_KEYS, "hist": self.online_hist, "approximate": True,
                               "bin_width": width}
        if self.online_rps is not None:
            out["rps"] = self.online_rps.to(torch.float32)
        return out

    def save_summary(self, path: str, *, hist_bins: int = 256, hist_max: float | None = None,
                     series: bool = True) -> dict[str, np.ndarray]:
        """Columnar dump of the sweep: one row per scenario with its seed, parameter
        columns, counts, flags, the 8 latency statistics, the 1-s RPS series, a latency histogram
        and the mean / maximum of every sampled series.  ``.npz`` (numpy) or ``.parquet`` (pyarrow;
        array-valued columns become list columns).  Returns the columns."""
        summ_kwargs: dict[str, Any] = {"rps": True, "series": series and self._samples_t is not None}
        stats0 = None
        if hist_bins:
            if hist_max is None:      # 1.25 x the largest latency of the sweep
                stats0 = self.summary(rps=False)["stats"].cpu().numpy()
                mx = np.nanmax(stats0[:, 7]) if np.isfinite(stats0[:, 7]).any() else 1.0
                hist_max = float(mx) * 1.25 or 1.0
            summ_kwargs.update(hist_bins=hist_bins, hist_max=hist_max)
        summ = self.summary(**summ_kwargs)
        cols: dict[str, np.ndarray] = {"seed": np.asarray(self.seeds, dtype=np.uint64)}
        for k, v in self.overrides.items():
            cols[f"param:{k}"] = np.asarray(v, dtype=np.float64)
        for name, slot in (("generated", _abi.CNT_GENERATED), ("completed", _abi.CNT_COMPLETED),
                           ("dropped", _abi.CNT_DROPPED), ("request_events", _abi.CNT_EVENTS),
                           ("ticks", _abi.CNT_TICKS), ("flags", _abi.CNT_FLAGS)):
            # (counts[CNT_MAX_LIVE] is a diagnostic -- the next-event kernels' high-water mark of live requests; the stage-parallel
            # kernel writes 0, or for general servers its rounds solved at once << 16 | walked event by event, each half
            # saturating at 65 535 -- and is not part of the on-disk summary: one sweep may mix both paths)
            cols[name] = self.counts[:, slot].copy()
        stats = summ["stats"].cpu().numpy()
        for j, k in enumerate(LATENCY_KEYS):
            cols[f"latency:{k}"] = stats[:, j].copy()
        if "rps" in summ:
            cols["rps"] = summ["rps"].cpu().numpy()
        if "hist" in summ:
            cols["latency_hist"] = summ["hist"].cpu().numpy().view(np.uint32)
            cols["latency_hist_edges"] = np.linspace(0.0, float(hist_max), hist_bins + 1)
        if "series_mean" in summ:
            cols["series_mean"] = summ["series_mean"].cpu().numpy()
            cols["series_max"] = self.decode_series_max(summ["series_max"].cpu().numpy())
            cols["series_names"] = np.asarray(self.series_names())
        path = str(path)
        if path.endswith(".parquet"):
            import pyarrow as pa
            import pyarrow.parquet as pq

            n = len(self)
            table = {k: (pa.array(list(v)) if v.ndim == 2 and v.shape[0] == n else pa.array(v))
                     for k, v in cols.items() if v.shape[:1] == (n,)}
            meta = {k: ",".join(map(str, v.tolist())) for k, v in cols.items() if v.shape[:1] != (n,)}
            pq.write_table(pa.table(table).replace_schema_metadata(meta), path)
        else:
            np.savez_compressed(path, **cols)
        return cols

    def decode_series_max(self, words: np.ndarray) -> np.ndarray:
        """``series_max`` words [n, n_series] -> float64 values (counts as they are, the ram_in_use
        columns decoded from their float32 bits)."""
        w = np.ascontiguousarray(words).view(np.uint32)
        out = w.astype(np.float64)
        j = np.arange(w.shape[1])
        ram = (j >= self.plan.n_edges) & ((j - self.plan.n_edges) % 3 == 2)
        out[:, ram] = w[:, ram].view(np.float32).astype(np.float64)
        return out

    def series_names(self) -> list[str]:
        """Names of the sampled series in device order: edges, then ready/io/ram per server."""
        names = [f"{e}:edge_concurrent_connection" for e in self.plan.edge_ids]
        for sid in self.plan.server_ids:
            names += [f"{sid}:ready_queue_len", f"{sid}:event_loop_io_sleep", f"{sid}:ram_in_use"]
        return names

    def aggregate(self, level: float = 0.95) -> dict[str, Any]:
        """Monte-Carlo aggregation over the scenarios of the sweep (the reference's roadmap
        item, ROADMAP.md:23-29): mean, standard deviation and normal-approximation confidence
        half-width of every latency statistic, plus the mean RPS band (5th/95th percentile
        across scenarios per 1-s window)."""
        return aggregate_summary(self.summary(rps=True), level)

    def differing_scenarios(self, other: "BatchedResults", chunk: int = 512) -> np.ndarray:
        """Indices of the scenarios whose results differ from ``other``'s, compared ON THE DEVICE over the whole batch
        (see :func:`differing_scenarios`): two runs of one sweep by different kernel families must return an empty array."""
        return differing_scenarios(self._counts_t, self._clock_t, self._samples_t,
                                   other._counts_t, other._clock_t, other._samples_t, chunk)  # noqa: SLF001


def differing_scenarios(counts_a: Any, clock_a: Any, samples_a: Any, counts_b: Any, clock_b: Any, samples_b: Any,
                        chunk: int = 512) -> np.ndarray:
    """Whole-batch comparison of two result sets of one sweep without leaving HBM (10 000 LB-2 scenarios at T = 600 s
    are 2 x 18 GB): the counts (generated, completed, dropped, request-events, ticks, flags, timeline marks), every
    ``rqs_clock`` row a scenario completed (client.py:62-69) as BIT PATTERNS, and every sample word of every tick the
    scenario reached (collector.py:50-66).  Rows behind a scenario's own counts are uninitialised memory and not looked
    at.  Returns the differing scenario indices (sorted int64 array; empty = bit-identical)."""
    import torch

    n = int(counts_a.shape[0])
    if int(counts_b.shape[0]) != n:
        msg = f"batches of {n} and {int(counts_b.shape[0])} scenarios"
        raise ValueError(msg)
    ca, cb = counts_a.to(torch.int64) & 0xFFFFFFFF, counts_b.to(torch.int64) & 0xFFFFFFFF
    cols = [_abi.CNT_GENERATED, _abi.CNT_COMPLETED, _abi.CNT_DROPPED, _abi.CNT_EVENTS, _abi.CNT_TICKS, _abi.CNT_FLAGS, _abi.CNT_MARKS]
    bad = (ca[:, cols] != cb[:, cols]).any(dim=1)
    if (clock_a is None) != (clock_b is None) or (samples_a is None) != (samples_b is None):
        msg = "one batch kept an output the other did not"
        raise ValueError(msg)
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        if clock_a is not None:
            cap = min(int(clock_a.shape[1]), int(clock_b.shape[1]))
            done = ca[lo:hi, _abi.CNT_COMPLETED].clamp(max=cap)
            live = torch.arange(cap, device=done.device)[None, :] < done[:, None]
            x = clock_a[lo:hi, :cap].view(torch.int64)
            y = clock_b[lo:hi, :cap].view(torch.int64)
            bad[lo:hi] |= ((x != y).any(dim=2) & live).any(dim=1)
        if samples_a is not None:
            cap = min(int(samples_a.shape[1]), int(samples_b.shape[1]))
            pitch = min(int(samples_a.shape[2]), int(samples_b.shape[2]))
            ticks = ca[lo:hi, _abi.CNT_TICKS].clamp(max=cap)
            live = torch.arange(cap, device=ticks.device)[None, :] < ticks[:, None]
            bad[lo:hi] |= ((samples_a[lo:hi, :cap, :pitch] != samples_b[lo:hi, :cap, :pitch]).any(dim=2) & live).any(dim=1)
    return torch.nonzero(bad).reshape(-1).cpu().numpy()


def aggregate_summary(summ: dict[str, Any], level: float = 0.95) -> dict[str, Any]:
    """Monte-Carlo aggregation of a ``summary()`` dict (see :meth:`BatchedResults.aggregate`)."""
    from statistics import NormalDist

    import torch

    st = summ["stats"]                                  # [n, 8] on the run's device: reduced there,
    ok = st[:, 0] > 0                                   # only the 8-vectors and the [T] bands come back
    z = NormalDist().inv_cdf(0.5 + level / 2.0)
    k = int(ok.sum())
    body = st[ok]
    nan8 = np.full(8, np.nan)
    mean = body.mean(dim=0).cpu().numpy() if k else nan8
    sd = body.std(dim=0, unbiased=True).cpu().numpy() if k > 1 else nan8
    out: dict[str, Any] = {
        "n": k,
        "keys": LATENCY_KEYS,
        "mean": dict(zip(LATENCY_KEYS, mean.tolist())),
        "std": dict(zip(LATENCY_KEYS, sd.tolist())),
        "ci_halfwidth": dict(zip(LATENCY_KEYS, (z * sd / np.sqrt(max(k, 1))).tolist())),
        "level": level,
    }
    if "rps" in summ:
        r = summ["rps"].to(torch.float64)
        out["rps_mean"] = r.mean(dim=0).cpu().numpy()
        q = torch.quantile(r, torch.tensor([0.05, 0.95], dtype=torch.float64, device=r.device), dim=0)
        out["rps_p05"], out["rps_p95"] = q[0].cpu().numpy(), q[1].cpu().numpy()
    return out


class ShardedResults:
    """A sweep run on several devices by ONE process (``SimulationRunner(devices=[...])``): the shards'
    :class:`BatchedResults` behind the indices of the original sweep."""

    def __init__(self, shards: list[BatchedResults], index: list[np.ndarray], wall_s: float) -> None:
        self.shards, self.index, self.wall_s = shards, index, wall_s
        n = sum(len(ix) for ix in index)
        self._where = np.zeros((n, 2), dtype=np.int64)
        for k, ix in enumerate(index):
            self._where[ix, 0] = k
            self._where[ix, 1] = np.arange(len(ix))
        self.plan = shards[0].plan
        self.counts = np.zeros((n, _abi.CNT_SLOTS), dtype=np.uint32)
        self.seeds = np.zeros(n, dtype=np.uint64)
        for k, ix in enumerate(index):
            self.counts[ix] = shards[k].counts
            self.seeds[ix] = shards[k].seeds
        self.kernel_ms = max(s.kernel_ms for s in shards)
        self.flow_reason = shards[0].flow_reason
        #: per-scenario parameter columns, in the order of the original sweep
        self.overrides: dict[str, np.ndarray] = {}
        for key in shards[0].overrides:
            col = np.zeros(n, dtype=np.float64)
            for k, ix in enumerate(index):
                col[ix] = shards[k].overrides[key]
            self.overrides[key] = col

    @property
    def engine_stats(self) -> list[_abi.AfStats]:
        """One ``af_stats_t`` per shard (per device), in shard order."""
        return [s.engine_stats for s in self.shards]

    def series_names(self) -> list[str]:
        return self.shards[0].series_names()

    def decode_series_max(self, words: np.ndarray) -> np.ndarray:
        return self.shards[0].decode_series_max(words)

    def save_summary(self, path: str, **kw: Any) -> dict[str, np.ndarray]:
        """:meth:`BatchedResults.save_summary` over every shard: the shards' columns are merged behind the indices of
        the original sweep and written once (``.npz`` / ``.parquet``)."""
        import tempfile

        if kw.get("hist_bins", 256) and kw.get("hist_max") is None:      # one histogram range for the whole sweep
            mx = 0.0
            for s in self.shards:
                st = s.summary(rps=False)["stats"].cpu().numpy()[:, 7]
                mx = max(mx, float(np.nanmax(st)) if np.isfinite(st).any() else 0.0)
            kw["hist_max"] = mx * 1.25 or 1.0
        n = len(self)
        cols: dict[str, np.ndarray] = {}
        with tempfile.TemporaryDirectory() as tmp:
            for k, (s, ix) in enumerate(zip(self.shards, self.index)):
                part = s.save_summary(f"{tmp}/shard{k}.npz", **kw)
                for name, v in part.items():
                    if v.shape[:1] == (len(ix),) and name not in ("latency_hist_edges", "series_names"):
                        if name not in cols:
                            cols[name] = np.zeros((n, *v.shape[1:]), dtype=v.dtype)
                        cols[name][ix] = v
                    else:
                        cols[name] = v
        path = str(path)
        if path.endswith(".parquet"):
            import pyarrow as pa
            import pyarrow.parquet as pq

            table = {k: (pa.array(list(v)) if v.ndim == 2 and v.shape[0] == n else pa.array(v))
                     for k, v in cols.items() if v.shape[:1] == (n,)}
            meta = {k: ",".join(map(str, v.tolist())) for k, v in cols.items() if v.shape[:1] != (n,)}
            pq.write_table(pa.table(table).replace_schema_metadata(meta), path)
        else:
            np.savez_compressed(path, **cols)
        return cols

    def __len__(self) -> int:
        return int(self.counts.shape[0])

    def __getitem__(self, i: int) -> ScenarioResults:
        k, j = self._where[int(i)]
        return self.shards[int(k)][int(j)]

    def __iter__(self) -> Iterator[ScenarioResults]:
        return (self[i] for i in range(len(self)))

    @property
    def flags(self) -> np.ndarray:
        return self.counts[:, _abi.CNT_FLAGS]

    @property
    def request_events(self) -> np.ndarray:
        return self.counts[:, _abi.CNT_EVENTS].astype(np.int64)

    def raise_on_overflow(self) -> None:
        for s in self.shards:
            s.raise_on_overflow()

    def raise_on_negative_delay(self) -> None:
        for s in self.shards:
            s.raise_on_negative_delay()

    def summary(self, **kw: Any) -> dict[str, Any]:
        """Per-scenario summaries of every shard (each computed on its own device), concatenated on the
        first shard's device in the order of the original sweep."""
        import torch

        parts = [s.summary(**kw) for s in self.shards]
        dev = parts[0]["stats"].device
        order = torch.as_tensor(np.argsort(np.concatenate(self.index), kind="stable"), device=dev)
        out: dict[str, Any] = {"keys": LATENCY_KEYS}
        for key in ("stats", "rps", "hist", "series_mean", "series_max"):
            if key in parts[0]:
                out[key] = torch.cat([p[key].to(dev) for p in parts], dim=0).index_select(0, order)
        return out

    def aggregate(self, level: float = 0.95) -> dict[str, Any]:
        return aggregate_summary(self.summary(rps=True), level)


def load_summary(path: str) -> dict[str, np.ndarray]:
    """Read a sweep summary written by :meth:`BatchedResults.save_summary` (``.npz`` or ``.parquet``)
    back into the same columns: one row per scenario (``seed``, ``param:*``, counts, ``latency:*``,
    ``rps`` [n, floor(T)], ``latency_hist`` [n, bins], ``series_mean`` / ``series_max`` [n, n_series])
    plus the per-sweep vectors (``latency_hist_edges``, ``series_names``)."""
    path = str(path)
    if path.endswith(".parquet"):
        import pyarrow.parquet as pq

        table = pq.read_table(path)
        cols: dict[str, np.ndarray] = {}
        for name in table.column_names:
            col = table.column(name).to_pylist()
            cols[name] = np.asarray(col, dtype=np.uint64 if name == "seed" else None)
        for k, v in (table.schema.metadata or {}).items():
            key, text = k.decode(), v.decode()
            if key == "series_names":
                cols[key] = np.asarray(text.split(","))
            else:
                cols[key] = np.asarray([float(x) for x in text.split(",")] if text else [], dtype=np.float64)
        return cols
    with np.load(path, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}
