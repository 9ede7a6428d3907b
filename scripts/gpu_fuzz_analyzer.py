"""GPU fuzz of the batched analyzer (af_summary_kernel / af_series_kernel, SURVEY 8 a11 + f1): random payloads of every family, six
scenarios each -- the 8 latency statistics, the 1-s RPS series, a latency histogram with a random range and the per-series mean / max
of `BatchedResults.summary()` against oracle/analyzer_oracle.py (numpy's own order statistics) on the downloaded outputs; then the
kernel-side summary (`online_summary=`: histogram + 1-s completion counts written by the simulation kernels, with and without the
per-request clock) against the same functions.

    python scripts/gpu_fuzz_analyzer.py [payloads, default 200] [first payload index, default 0]

Every number bit-exact -- order statistics, RPS, histogram, series, and since round 6 mean / std too (numpy's own summation order).
One JSON line."""
import json
import random
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from asyncflow_amd.runner import SimulationRunner  # noqa: E402
from oracle import analyzer_oracle as ao  # noqa: E402
from oracle.scenarios import flow_payload, gateway_lb, random_payload, server_tiers, tie_storm  # noqa: E402

n_payloads = int(sys.argv[1]) if len(sys.argv) > 1 else 200
k0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
N = 6
EXACT = [0, 2, 4, 5, 6, 7]   # total, median, p95, p99, min, max


def make(k: int) -> dict:
    rng = random.Random(55000 + k)
    kind = k % 5
    if kind == 0:
        return random_payload(rng, horizon=8)
    if kind == 1:
        return server_tiers(rng, horizon=10)
    if kind == 2:
        return tie_storm(rng, horizon=8)
    if kind == 3:
        return gateway_lb(front=rng.choice((1, 2)), users=rng.choice((5, 60, 300)), horizon=10, general=rng.random() < 0.3)
    p = flow_payload(rng, horizon=6)
    if rng.random() < 0.3:   # nearly nothing completes: the empty / one-element edge cases of the order statistics
        p["rqs_input"]["avg_active_users"]["mean"] = rng.choice((1, 2))
        p["rqs_input"]["avg_request_per_minute_per_user"]["mean"] = rng.choice((1, 10, 20))
    return p


t = {"payloads": 0, "scenarios": 0, "empty_scenarios": 0, "overflow_raised": 0, "online_checks": 0}
failures: list[str] = []
for k in range(k0, k0 + n_payloads):
    payload = make(k)
    rng = random.Random(k)
    seeds = np.arange(N, dtype=np.uint64) + 100 * k + 3
    try:
        res = SimulationRunner(simulation_input=payload, seeds=seeds, on_negative_delay="flag").run()
    except OverflowError:
        t["overflow_raised"] += 1
        continue
    bins = rng.choice((16, 64, 256))
    hmax = rng.choice((0.004, 0.05, 0.5, 4.0))
    summ = res.summary(rps=True, hist_bins=bins, hist_max=hmax, series=True)
    stats = summ["stats"].cpu().numpy()
    T = int(res.plan.total_time)
    rps = summ["rps"].cpu().numpy() if T > 0 else None
    hist = summ["hist"].cpu().numpy().view(np.uint32)
    smean = summ["series_mean"].cpu().numpy()
    smax = summ["series_max"].cpu().numpy().view(np.uint32)
    t["payloads"] += 1
    try:
        for i in range(N):
            sc = res[i]
            want = ao.latency_stats(sc.rqs_clock)
            t["scenarios"] += 1
            t["empty_scenarios"] += int(want[0] == 0)
            assert np.array_equal(stats[i][EXACT].view(np.uint64), want[EXACT].view(np.uint64)), (k, i, "order statistics", stats[i].tolist(), want.tolist())
            if want[0] > 0:      # (round 6: mean and std_dev bit-equal to numpy's too)
                assert stats[i][1] == want[1], (k, i, "mean", stats[i][1], want[1])
                assert stats[i][3] == want[3], (k, i, "std", stats[i][3], want[3])
            else:
                assert np.isnan(stats[i][1:]).all(), (k, i, "empty", stats[i].tolist())
            if rps is not None:
                assert np.array_equal(rps[i].astype(np.float64), ao.throughput_series(sc.rqs_clock, T)[1]), (k, i, "rps")
            assert np.array_equal(hist[i], ao.latency_histogram(sc.rqs_clock, bins, hmax)), (k, i, "histogram", bins, hmax)
            m, x = ao.series_mean_max(sc._samples, res.plan.n_edges)  # noqa: SLF001
            assert np.array_equal(smean[i].view(np.uint64), m.view(np.uint64)), (k, i, "series mean")
            assert np.array_equal(smax[i], x), (k, i, "series max")
            one = sc.get_latency_stats()
            if want[0] > 0:
                assert one["p95"] == stats[i][4] and one["total_requests"] == stats[i][0], (k, i, "accessor")
    except AssertionError as exc:
        failures.append(str(exc)[:400])
        print(f"DIFFERENT payload {k}: {str(exc)[:400]}", file=sys.stderr)
    # the kernel-side summary (online_summary: histogram + 1-s completion counts written by the simulation kernels themselves, with
    # and without the per-request clock) against the same oracle functions on the clock of the run above
    try:
        obins = rng.choice((64, 1024, 4096))
        omax = rng.choice((0.016, 0.256, 2.0))
        both = SimulationRunner(simulation_input=payload, seeds=seeds, on_negative_delay="flag", online_summary={"hist_bins": obins, "hist_max": omax}).run()
        lean = SimulationRunner(simulation_input=payload, seeds=seeds, on_negative_delay="flag", collect_clock=False, collect_samples=False,
                                online_summary={"hist_bins": obins, "hist_max": omax}).run()
        oh = both.online_hist.cpu().numpy().view(np.uint32)
        orps = both.online_rps.cpu().numpy() if T > 0 else None
        assert np.array_equal(both.counts[:, :5], res.counts[:, :5]), (k, "online: counts")
        assert np.array_equal(lean.counts[:, :5], res.counts[:, :5]), (k, "online, no outputs: counts")
        assert np.array_equal(lean.online_hist.cpu().numpy(), both.online_hist.cpu().numpy()), (k, "online, no outputs: histogram")
        if orps is not None:
            assert np.array_equal(lean.online_rps.cpu().numpy(), both.online_rps.cpu().numpy()), (k, "online, no outputs: rps")
        for i in range(N):
            ck = res[i].rqs_clock
            assert np.array_equal(oh[i], ao.latency_histogram(ck, obins, omax)), (k, i, "online histogram", obins, omax)
            if orps is not None:
                assert np.array_equal(orps[i].astype(np.float64), ao.throughput_series(ck, T)[1]), (k, i, "online rps")
        t["online_checks"] += N
        both.close()
        lean.close()
    except OverflowError:
        t["overflow_raised"] += 1
    except AssertionError as exc:
        failures.append(str(exc)[:400])
        print(f"DIFFERENT payload {k}: {str(exc)[:400]}", file=sys.stderr)
    res.close()
t["different"] = len(failures)
t["failures"] = failures[:10]
print(json.dumps(t))
