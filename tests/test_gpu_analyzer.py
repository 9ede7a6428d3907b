"""The HIP analyzer (af_engine_summarize, through the C ABI) vs the analyzer oracle on a real MI355X.

Bar: EVERY number bit-exact -- total, mean, median, std_dev, p95, p99, min, max, RPS windows, histogram, series
mean / max.  (Rounds 2 - 5 held mean and std_dev to 1e-12: the kernel summed in its own tree order.  Since round 6 it adds
in numpy's order -- pieces of 8 192, pairwise inside, eight running sums per leaf: af_summary.hpp -- and computes the
variance in numpy's two passes.)
"""

from __future__ import annotations

import numpy as np
import pytest

from asyncflow_amd import _abi
from asyncflow_amd.plan import lower
from oracle import analyzer_oracle as ao
from oracle.scenarios import lb_two_servers, lb_with_events, single_server

pytestmark = pytest.mark.gpu

def _check_stats(got: np.ndarray, want: np.ndarray, what=""):
    if want[0] > 0:
        assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), (what, got, want, got - want)
    else:
        assert got[0] == 0 and np.isnan(got[1:]).all()


def _summarize_synthetic(lats_list, starts_list=None, total_time=50, hist_bins=0, hist_max=0.0, cap_limit=None,
                         random_starts=()):
    """Feed hand-made rqs_clock rows straight to af_engine_summarize."""
    import torch

    from asyncflow_amd.engine import Engine

    plan = lower(single_server(horizon=total_time))
    n = len(lats_list)
    cap = max(max((len(x) for x in lats_list), default=1), 1)
    if cap_limit:
        cap = cap_limit
    rng = np.random.default_rng(7)
    clock = np.full((n, cap, 2), np.nan)
    counts = np.zeros((n, _abi.CNT_SLOTS), dtype=np.uint32)
    clocks = []
    for i, lat in enumerate(lats_list):
        lat = np.asarray(lat, dtype=np.float64)
        if starts_list is not None:
            start = np.asarray(starts_list[i], dtype=np.float64)
        elif i in random_starts:                               # finish - start then differs from lat by rounding
            start = rng.uniform(0.0, total_time - 1.0, size=lat.size)
        else:                                                   # finish - 0 == lat exactly: ties stay ties
            start = np.zeros(lat.size)
        fin = start + lat
        rows = np.stack([start, fin], axis=1) if lat.size else np.zeros((0, 2))
        counts[i, _abi.CNT_COMPLETED] = lat.size           # may exceed cap: the kernel clamps
        keep = rows[:cap]
        clock[i, : keep.shape[0]] = keep
        clocks.append(keep)
    dev = torch.device("cuda", 0)
    clock_t = torch.as_tensor(clock, device=dev)
    counts_t = torch.as_tensor(counts.view(np.int32), device=dev)
    stats = torch.empty((n, 8), dtype=torch.float64, device=dev)
    rps = torch.empty((n, total_time), dtype=torch.float32, device=dev)
    hist = torch.empty((n, max(hist_bins, 1)), dtype=torch.int32, device=dev)
    eng = Engine(plan, 0)
    eng.summarize(n, clock_ptr=clock_t.data_ptr(), clock_capacity=cap, samples_ptr=0, tick_capacity=0,
                  counts_ptr=counts_t.data_ptr(), stats_ptr=stats.data_ptr(), rps_ptr=rps.data_ptr(),
                  rps_buckets=total_time, hist_ptr=hist.data_ptr() if hist_bins else 0, hist_bins=hist_bins,
                  hist_max=hist_max)
    eng.close()
    return clocks, stats.cpu().numpy(), rps.cpu().numpy(), hist.cpu().numpy().view(np.uint32)


@pytest.fixture(params=["eight_waves", "four_waves"])
def last_pass(request, monkeypatch):
    """The forms of the analyzer's latency kernel: compiled for eight waves per SIMD (four scenarios per CU, the default since
    round 5) or for four (AF_SUMMARY_WPE=4: `af_summary_kernel<4>`, eight loads in flight per thread instead of four)."""
    if request.param == "four_waves":
        monkeypatch.setenv("AF_SUMMARY_WPE", "4")
    else:
        monkeypatch.delenv("AF_SUMMARY_WPE", raising=False)
    return request.param


def test_order_statistics_are_exact_on_adversarial_latency_sets(last_pass):
    rng = np.random.default_rng(2026)
    cases = [
        [],                                                     # no completion at all
        [0.25],                                                 # n = 1
        [0.5, 0.125],                                           # n = 2
        [3.0, 1.0, 2.0],
        rng.exponential(0.02, 511), rng.exponential(0.02, 512), rng.exponential(0.02, 513),
        rng.exponential(0.02, 20_001),
        np.full(5_000, 0.0625),                                 # every latency equal: radix runs to the last bit
        rng.choice([0.001, 0.002, 0.004], 30_000),              # three values, > kCand ties each
        np.concatenate([np.full(4_000, 0.01), 0.01 + rng.uniform(0, 1e-15, 4_000)]),   # split only by the last mantissa bits
        10.0 ** rng.uniform(-300, 300, 10_000),                 # the whole exponent range
        np.concatenate([np.zeros(700), rng.uniform(0, 1, 701)]),  # many exact zeros
        rng.lognormal(-4.0, 0.5, 75_861),                       # LB-2 sized
        np.arange(1, 101) * 0.001,                              # n = 100: p95 / p99 interpolate with t < 0.5 and t >= 0.5
        np.arange(1, 34) * 0.5,
        np.concatenate([rng.lognormal(-4.0, 0.3, 60_000), rng.lognormal(2.0, 0.3, 4_000)]),   # p95 / p99 in binades the first 512 may miss
        1.0 + rng.uniform(0, 1e-9, 50_000),                    # deviations 1e-9 of the mean
    ]
    # numpy's summation order (mean, std_dev): below / at / above its 8-element rows, 128-element leaves and 8 192-element pieces
    cases += [rng.exponential(0.02, n) for n in (5, 7, 8, 9, 15, 16, 127, 128, 129, 130, 135, 136, 143, 144, 257, 1_000, 4_097,
                                                  8_185, 8_191, 8_192, 8_193, 8_199, 8_200, 8_192 + 129, 16_384, 16_385,
                                                  3 * 8_192 + 64, 5 * 8_192 - 1, 132_096, 200_001)]
    clocks, stats, rps, _ = _summarize_synthetic(cases, random_starts=(4, 5, 6, 7, 13, *range(18, 48, 3)))
    for i, ck in enumerate(clocks):
        _check_stats(stats[i], ao.latency_stats(ck), f"case {i}")
        assert np.array_equal(rps[i].astype(np.float64), ao.throughput_series(ck, 50)[1]), f"rps case {i}"


def test_window_edges_histogram_and_capacity_clamp(last_pass):
    starts = [np.zeros(6)]
    lats = [np.array([0.0, 1.0, 1.0000000000000002, 2.0, 49.99, 50.0])]
    clocks, stats, rps, hist = _summarize_synthetic(lats, starts, hist_bins=8, hist_max=4.0)
    want = ao.throughput_series(clocks[0], 50)[1]
    assert np.array_equal(rps[0].astype(np.float64), want) and want[0] == 2.0 and want[49] == 2.0
    assert np.array_equal(hist[0], ao.latency_histogram(clocks[0], 8, 4.0))
    # counts says 1000 completions, the clock buffer holds 300: only the stored rows are analysed
    rng = np.random.default_rng(5)
    clocks, stats, _, _ = _summarize_synthetic([rng.exponential(0.03, 1000)], cap_limit=300, random_starts=(0,))
    assert clocks[0].shape[0] == 300
    _check_stats(stats[0], ao.latency_stats(clocks[0]))


@pytest.mark.parametrize("payload_fn", [lambda: lb_two_servers(horizon=30), lambda: lb_with_events(users=150, horizon=40, scale=0.05)])
def test_summary_of_a_simulated_batch_matches_the_oracle(payload_fn, last_pass):
    from asyncflow_amd.runner import SimulationRunner

    payload = payload_fn()
    seeds = 0x5EED0000 + np.arange(40, dtype=np.uint64)
    res = SimulationRunner(simulation_input=payload, seeds=seeds).run()
    summ = res.summary(rps=True, hist_bins=64, hist_max=0.128, series=True)
    stats = summ["stats"].cpu().numpy()
    rps = summ["rps"].cpu().numpy()
    hist = summ["hist"].cpu().numpy().view(np.uint32)
    smean = summ["series_mean"].cpu().numpy()
    smax = summ["series_max"].cpu().numpy().view(np.uint32)
    T = int(res.plan.total_time)
    for i in range(len(res)):
        sc = res[i]
        _check_stats(stats[i], ao.latency_stats(sc.rqs_clock), f"scenario {i}")
        assert np.array_equal(rps[i].astype(np.float64), ao.throughput_series(sc.rqs_clock, T)[1])
        assert np.array_equal(hist[i], ao.latency_histogram(sc.rqs_clock, 64, 0.128))
        m, x = ao.series_mean_max(sc._samples, res.plan.n_edges)  # noqa: SLF001
        assert np.array_equal(smean[i].view(np.uint64), m.view(np.uint64)) and np.array_equal(smax[i], x)
        # ram_in_use columns hold float32 values: mean / max of the list the drop-in accessor returns
        sm = sc.get_sampled_metrics()
        for v, sid in enumerate(res.plan.server_ids):
            col = res.plan.n_edges + 3 * v + 2
            ram = np.asarray(sm["ram_in_use"][sid], dtype=np.float64)
            assert ram.max() > 0.0 and smean[i][col] == ram.sum() / len(ram)
            assert res.decode_series_max(smax[i:i + 1])[0, col] == ram.max()
            assert smean[i][col - 1] == np.mean(sm["event_loop_io_sleep"][sid])
        # the drop-in accessor of one scenario agrees with the batched row
        one = sc.get_latency_stats()
        assert one["p95"] == stats[i][4] and one["total_requests"] == stats[i][0]
    agg = res.aggregate()
    assert agg["n"] == 40 and agg["ci_halfwidth"]["p95"] > 0.0
    assert abs(agg["mean"]["p95"] - stats[:, 4].mean()) < 1e-15
    assert agg["rps_mean"].shape == (T,)


def test_grid_sweep_and_on_disk_summary(tmp_path):
    from asyncflow_amd import SimulationRunner, expand_grid

    users = "rqs_input.avg_active_users.mean"
    rtt = "topology_graph.edges[*].latency.mean"
    sw = expand_grid({users: [20, 200, 60], rtt: [0.001, 0.01]}, replicas=3, seed_base=7000, order_by_load=users)
    payload = lb_two_servers(horizon=20)
    res = SimulationRunner(simulation_input=payload, **sw.runner_kwargs()).run()
    cols = res.save_summary(str(tmp_path / "sweep.npz"))
    z = np.load(tmp_path / "sweep.npz")
    assert z["seed"].tolist() == sw.seeds.tolist() and np.array_equal(z[f"param:{users}"], sw.columns[users])
    assert z["rps"].shape == (18, 20) and z["latency_hist"].shape == (18, 256)
    assert np.array_equal(z["latency_hist"].sum(axis=1), z["completed"])
    assert list(z["series_names"])[:1] == [f"{res.plan.edge_ids[0]}:edge_concurrent_connection"]
    p95 = sw.by_point(cols["latency:p95"])                     # [users, rtt, replica]
    assert p95.shape == (3, 2, 3) and (p95[:, 1].mean() > p95[:, 0].mean())     # 10x the per-hop latency
    for i in (0, 9, 17):
        _check_stats(np.array([cols[f"latency:{k}"][i] for k in ao.LATENCY_KEYS]), ao.latency_stats(res[i].rqs_clock))
    res.save_summary(str(tmp_path / "sweep.parquet"))
    import pyarrow.parquet as pq

    t = pq.read_table(tmp_path / "sweep.parquet")
    assert t.num_rows == 18 and t.column("latency:p95").to_pylist() == cols["latency:p95"].tolist()
    assert len(t.column("rps")[0].as_py()) == 20
    # load side: both formats come back as the columns save_summary() wrote
    from asyncflow_amd.results import load_summary

    for name in ("sweep.npz", "sweep.parquet"):
        back = load_summary(str(tmp_path / name))
        assert set(back) == set(cols), name
        for k, v in cols.items():
            if v.dtype.kind in "US":
                assert list(back[k]) == list(v), (name, k)
            else:
                assert np.array_equal(np.asarray(back[k], dtype=np.float64), v.astype(np.float64), equal_nan=True), (name, k)
    ram_col = res.plan.n_edges + 2
    assert cols["series_max"][:, ram_col].max() in (128.0 * np.arange(1, 17))      # decoded MB, not float bits


def test_kernel_side_summary_without_the_per_request_clock():
    """SimulationRunner(online_summary=...): histogram + 1-s windows accumulated by the next-event kernel
    (exact counts), usable with collect_clock=False; statistics read from it are accurate to one bin."""
    from asyncflow_amd import SimulationRunner

    payload = lb_with_events(users=150, horizon=40, scale=0.05)
    seeds = np.arange(48, dtype=np.uint64) + 9
    bins, hist_max = 2048, 0.256
    both = SimulationRunner(simulation_input=payload, seeds=seeds, online_summary={"hist_bins": bins, "hist_max": hist_max}).run()
    hist = both.online_hist.cpu().numpy().view(np.uint32)
    rps = both.online_rps.cpu().numpy()
    T = int(both.plan.total_time)
    for i in range(len(both)):
        ck = both[i].rqs_clock
        assert np.array_equal(hist[i], ao.latency_histogram(ck, bins, hist_max))
        assert np.array_equal(rps[i].astype(np.float64), ao.throughput_series(ck, T)[1])
    exact = both.summary()["stats"].cpu().numpy()
    lean = SimulationRunner(simulation_input=payload, seeds=seeds, collect_clock=False, collect_samples=False,
                            online_summary={"hist_bins": bins, "hist_max": hist_max}).run()
    assert np.array_equal(lean.counts[:, :5], both.counts[:, :5])
    summ = lean.summary()
    approx = summ["stats"].cpu().numpy()
    assert summ["approximate"] and np.array_equal(approx[:, 0], exact[:, 0])
    width = hist_max / bins
    # numpy interpolates between two order statistics, which in a thin tail lie several bins apart
    assert np.abs(approx[:, 2] - exact[:, 2]).max() <= width                          # median
    assert np.abs(approx[:, [4, 5]] - exact[:, [4, 5]]).max() <= 4 * width            # p95, p99
    assert np.abs(approx[:, 1] - exact[:, 1]).max() <= width / 2 + 1e-12             # mean
    assert np.all(approx[:, 6] <= exact[:, 6]) and np.all(exact[:, 6] - approx[:, 6] <= width)
    assert np.all(approx[:, 7] >= exact[:, 7]) and np.all(approx[:, 7] - exact[:, 7] <= width)
    assert np.array_equal(summ["rps"].cpu().numpy(), rps.astype(np.float32))


def _summary_tensors_equal(a: dict, b: dict) -> None:
    import torch

    for key in ("stats", "rps", "hist", "series_mean", "series_max"):
        assert (key in a) == (key in b), key
        if key in a:
            x, y = a[key], b[key]
            if x.dtype.is_floating_point:       # bit patterns: NaN rows (no completion) compare equal too
                x, y = x.view(torch.int64 if x.dtype == torch.float64 else torch.int32), y.view(torch.int64 if y.dtype == torch.float64 else torch.int32)
            assert torch.equal(x, y), key


def test_run_and_analyzer_in_one_call_give_what_the_two_calls_give():
    """`af_engine_run_summarized` (round 6): the analyzer starts on a second stream once the scenarios of the stage-parallel kernel's
    full residency rounds have finished (a counter its waves bump; `hipStreamWaitValue32`) and runs beside the last, partial round;
    a workgroup that meets a scenario still being simulated (done flag clear) leaves it to a retry launch after the kernel.  Same
    sweep through `SimulationRunner(summary=...)` (one call) and through `run()` + `summary()` (two): every summary tensor
    bit-equal; with 10 000 replicas the overlap is really taken (~8 192 scenarios = two rounds of 4 096 resident waves)."""
    from asyncflow_amd.runner import SimulationRunner
    from oracle.scenarios import tie_storm
    import random

    kw = dict(rps=True, hist_bins=64, hist_max=0.25, series=True)
    payload = lb_two_servers(horizon=30)
    seeds = np.arange(10_000, dtype=np.uint64) + 77
    two = SimulationRunner(simulation_input=payload, seeds=seeds, specialise=True).run()
    one = SimulationRunner(simulation_input=payload, seeds=seeds, specialise=True, summary=kw).run()
    assert int(one.engine_stats.flow_scenarios) == 10_000 and int(one.engine_stats.flow_fallback) == 0
    assert 8_000 <= int(one.engine_stats.summary_overlapped) <= 10_000 and float(one.engine_stats.summary_beside_ms) > 0.0
    assert np.array_equal(one.counts, two.counts)
    assert one.summary(**kw)["stats"].data_ptr() == one._summary_from_run["stats"].data_ptr()          # noqa: SLF001  (no second launch)
    _summary_tensors_equal(one.summary(**kw), two.summary(**kw))
    i = 9_999                                                                                           # ... and against numpy
    assert np.array_equal(one.summary(**kw)["stats"][i].cpu().numpy().view(np.uint64), ao.latency_stats(two[i].rqs_clock).view(np.uint64))
    # the generic kernels (no plan-specialised build) split the same way
    gen = SimulationRunner(simulation_input=payload, seeds=seeds[:9_000], specialise=False, summary=kw).run()
    assert 4_000 <= int(gen.engine_stats.summary_overlapped) <= 9_000 and int(gen.engine_stats.specialised_launches) == 0
    _summary_tensors_equal(gen.summary(**kw), SimulationRunner(simulation_input=payload, seeds=seeds[:9_000], specialise=False).run().summary(**kw))
    # scenarios the kernel hands back are simulated again after the first part was analysed: the sweep is summarised again
    storm = tie_storm(random.Random(777_003), horizon=12)
    s_seeds = np.arange(6_000, dtype=np.uint64) + 5
    s_one = SimulationRunner(simulation_input=storm, seeds=s_seeds, summary=kw).run()
    s_two = SimulationRunner(simulation_input=storm, seeds=s_seeds).run()
    assert int(s_one.engine_stats.summary_overlapped) == 0 or int(s_one.engine_stats.flow_fallback) == 0
    _summary_tensors_equal(s_one.summary(**kw), s_two.summary(**kw))
    # a sweep over the load is launched heaviest first (an order array): whatever is finished when the gate opens, in any order
    users = np.linspace(20.0, 400.0, 6_000)
    g_one = SimulationRunner(simulation_input=payload, seeds=s_seeds, sweep={"rqs_input.avg_active_users.mean": users}, summary=kw).run()
    g_two = SimulationRunner(simulation_input=payload, seeds=s_seeds, sweep={"rqs_input.avg_active_users.mean": users}).run()
    _summary_tensors_equal(g_one.summary(**kw), g_two.summary(**kw))
