"""The BENCHED batches, compared in full.

`bench.py`'s own `RankSweep` -- the object the timed region steps -- runs each BASELINE workload at the size the bench
line is quoted on, once on the stage-parallel kernel (plan-specialised build, as benched) and once on the next-event
kernels, into two sets of HBM buffers; `asyncflow_amd.results.differing_scenarios` then compares, ON THE DEVICE,

* the counts of every scenario (generated, completed, dropped, request-events, ticks, flags, marks),
* every `rqs_clock` row every scenario completed, as bit patterns (client.py:62-69),
* every sample word of every tick (collector.py:50-66),

and >= 1 024 scenarios spread evenly over each batch are held to the CPU ORACLE itself (the checker pinned on the reference's
fixtures), run on every host core by `oracle/bulk.py`: counts, a 128-bit digest of the rqs_clock rows and one of the sampled
series per scenario.  Chain of custody: reference == oracle (tests/golden) ; oracle == BOTH kernel families on >= 1 024
scenarios of every benched batch (here; round 5: 32) ; next-event kernels == stage-parallel kernel over ALL scenarios of the
benched batch (here) -- a defect shared by the two kernel families (they share af_math.hpp, af_pregen.hpp, the plan lowering
and the output layout) no longer hides behind the device-side comparison.

Config 2: 10 000 seed replicas (2 x 18 GB of outputs); config 3: the 100 x 100 users x RTT grid; config 4: ALL 100 000
scenarios (round 5: every eighth), slice by slice; config 5: ALL 50 000 replicas (round 5: 4 096); config 6 (general servers,
not a BASELINE config): 2 048 replicas in tests/test_gpu_flow.py.
"""

from __future__ import annotations

import numpy as np
import pytest

from asyncflow_amd import _abi
from asyncflow_amd.results import differing_scenarios
from oracle import bulk

pytestmark = pytest.mark.gpu

BENCHED = {2: 10_000, 3: 10_000, 4: 100_000, 5: 50_000}


def _sweep(config: int, scenarios: int, extra: list[str]):
    import torch

    import bench

    args = bench.make_parser().parse_args(["--config", str(config), "--scenarios", str(scenarios), *extra])
    args.horizon = args.horizon or None
    wl = bench.build_workload(config, 0, 1, args.scenarios, args.horizon)
    sw = bench.RankSweep(wl, torch.device("cuda", 0), args)
    sw.prepare()
    return sw


def _oracle_bulk(sw, k: int, lo: int = 0, hi: int | None = None) -> int:
    """>= k scenarios spread evenly over [lo, hi) of the sweep -- whose outputs are resident in sw's buffers at rows
    i - lo -- against the oracle: counts, digest of every completed rqs_clock row, digest of every sample word."""
    import torch

    hi = sw.n if hi is None else hi
    m = hi - lo
    k = min(k, m)
    picks = sorted({lo + int(round(j * (m - 1) / max(k - 1, 1))) for j in range(k)})
    code_name = {v: name for name, v in _abi.PARAM_CODES.items()}
    overrides = [[(code_name[c], int(idx), float(col[i])) for c, idx, col, _ in sw.over] for i in picks]
    want = bulk.simulate_many(sw.plan.payload, [int(sw.seeds[i]) for i in picks], overrides, clock_capacity=sw.clock_cap)
    counts = sw.counts.cpu().numpy().view(np.uint32)
    n_series = sw.plan.n_series
    bad: list[tuple[int, str]] = []
    for g0 in range(0, len(picks), 64):          # 64 scenarios' outputs per copy (~0.1 GB)
        grp = picks[g0:g0 + 64]
        rows = torch.as_tensor([i - lo for i in grp], device=sw.clock.device)
        clock = sw.clock.index_select(0, rows).cpu().numpy()
        samples = sw.samples.index_select(0, rows)[:, :, :n_series].transpose(1, 2).contiguous().cpu().numpy().view(np.uint32)
        for j, i in enumerate(grp):
            w_counts, w_clock, w_samples, _ = want[g0 + j]
            got = counts[i]
            if got[:5].tolist() != w_counts[:5] or int(got[_abi.CNT_MARKS]) != w_counts[_abi.CNT_MARKS] \
                    or (int(got[_abi.CNT_FLAGS]) & 0xFF) != (w_counts[_abi.CNT_FLAGS] & 0xFF):
                bad.append((i, f"counts {got.tolist()} != {w_counts}"))
                continue
            done, ticks = int(got[_abi.CNT_COMPLETED]), int(got[_abi.CNT_TICKS])
            if bulk.digest_clock(clock[j, :done]) != w_clock:
                bad.append((i, "rqs_clock"))
            if bulk.digest_samples(samples[j, :, :ticks]) != w_samples:
                bad.append((i, "sampled series"))
    assert not bad, f"{len(bad)} of {len(picks)} scenarios differ from the oracle (first: {bad[:4]})"
    return len(picks)


def _both_families_and_the_oracle(config: int, oracle_k: int = 1024) -> None:
    """Every scenario of BASELINE config `config` at its benched size: stage-parallel kernel (as benched) == next-event
    kernels on the device, slice by slice, and >= oracle_k scenarios of the stage-parallel kernel's outputs == the oracle."""
    import torch

    flow = _sweep(config, 0, [])
    assert flow.n == BENCHED[config] and not flow.flow_reason
    # the next-event kernels over the same batch, a quarter of the output memory at a time (two full result sets of a
    # grid would be 2 x 80 GB next to the engine's own draw buffers); every slice is compared while it is resident
    seq = _sweep(config, 0, ["--no-flow", "--generic-kernels", "--hbm-budget-gb", "24"])
    assert seq.n == flow.n and np.array_equal(seq.seeds, flow.seeds) and seq.clock_cap == flow.clock_cap
    differ: list[int] = []
    checked = 0
    per_slice = -(-oracle_k // flow.n_slices)
    for lo in range(0, flow.n, flow.slice):
        hi = min(flow.n, lo + flow.slice)
        st = flow.run_slice(lo, hi)
        torch.cuda.synchronize()
        assert int(st.flow_scenarios) == hi - lo and int(st.flow_fallback) == 0                 # the benched launch ...
        assert int(st.specialised_launches) >= 1 and int(st.jit_fallbacks) == 0               # ... on its plan-specialised build
        checked += _oracle_bulk(flow, per_slice, lo, hi)
        for a in range(lo, hi, seq.slice):
            b = min(hi, a + seq.slice)
            st = seq.run_slice(a, b)
            torch.cuda.synchronize()
            assert int(st.flow_scenarios) == 0
            d = differing_scenarios(flow.counts[a:b], flow.clock[a - lo:b - lo], flow.samples[a - lo:b - lo],
                                    seq.counts[a:b], seq.clock[: b - a], seq.samples[: b - a])
            differ += (d + a).tolist()
    assert not differ, f"config {config}: {len(differ)} of {flow.n} scenarios differ between the kernel families (first: {differ[:8]})"
    assert checked >= oracle_k
    # the comparison looked at something: completions and ticks of the whole batch
    c = flow.counts.cpu().numpy().view(np.uint32)
    assert int(np.bitwise_or.reduce(c[:, _abi.CNT_FLAGS])) & _abi.FATAL_FLAGS == 0
    assert int(c[:, _abi.CNT_COMPLETED].astype(np.int64).sum()) > 1000 * flow.n and int(c[:, _abi.CNT_TICKS].min()) == 11_999
    if config == 4:
        assert int(c[:, _abi.CNT_MARKS].min()) > 0
    flow.eng.close()
    seq.eng.close()


@pytest.mark.parametrize("config", [2, 3])
def test_every_scenario_of_the_benched_batch_is_identical_on_both_kernel_families_and_1024_equal_the_oracle(config):
    _both_families_and_the_oracle(config)


def test_all_50000_replicas_of_config_5_are_identical_on_both_kernel_families_and_1024_equal_the_oracle():
    """BASELINE config 5 (8-server fan-out, log-normal ~1-s hops) at its benched size: two launches of the stage-parallel
    kernel, ~six passes of the next-event kernels (about a minute)."""
    _both_families_and_the_oracle(5)


def test_all_100000_scenarios_of_config_4_are_identical_on_both_kernel_families_and_1024_equal_the_oracle():
    """BASELINE config 4 (the users x RTT grid x 10 seeds with event_inj_lb.yml's spikes and outages) at its benched size:
    four launches of the stage-parallel kernel, the next-event kernels a quarter of the output memory at a time (about two
    minutes)."""
    _both_families_and_the_oracle(4)


def test_the_device_comparison_sees_one_flipped_bit():
    """The checker itself: one mantissa bit of one finish time, one sample word, one count."""
    import torch

    sw = _sweep(2, 64, ["--horizon", "20"])
    sw.step()
    torch.cuda.synchronize()
    counts, clock, samples = sw.counts.clone(), sw.clock.clone(), sw.samples.clone()
    assert differing_scenarios(sw.counts, sw.clock, sw.samples, counts, clock, samples).size == 0
    clock.view(torch.int64)[17, 5, 1] ^= 1
    samples[40, 100, 3] += 1
    counts[63, _abi.CNT_DROPPED] += 1
    assert differing_scenarios(sw.counts, sw.clock, sw.samples, counts, clock, samples).tolist() == [17, 40, 63]
    done = int(sw.counts[3, _abi.CNT_COMPLETED])
    clock[3, done:] = -1.0        # behind the scenario's own completions: not results
    assert differing_scenarios(sw.counts, sw.clock, sw.samples, counts, clock, samples).tolist() == [17, 40, 63]
    sw.eng.close()
