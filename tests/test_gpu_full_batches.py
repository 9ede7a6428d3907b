"""The BENCHED batches, compared in full.

`bench.py`'s own `RankSweep` -- the object the timed region steps -- runs each BASELINE workload at the size the bench
line is quoted on, once on the stage-parallel kernel (plan-specialised build, as benched) and once on the next-event
kernels, into two sets of HBM buffers; `asyncflow_amd.results.differing_scenarios` then compares, ON THE DEVICE,

* the counts of every scenario (generated, completed, dropped, request-events, ticks, flags, marks),
* every `rqs_clock` row every scenario completed, as bit patterns (client.py:62-69),
* every sample word of every tick (collector.py:50-66),

and >= 32 scenarios spread over the batch are also held to the CPU oracle (the checker pinned on the reference's
fixtures).  Chain of custody: reference == oracle (tests/golden) ; oracle == next-event kernels (test_gpu_parity.py) ;
next-event kernels == stage-parallel kernel over ALL scenarios of the benched batch (here).

Config 2: 10 000 seed replicas (2 x 18 GB of outputs); config 3: the 100 x 100 users x RTT grid; config 4: every eighth of
its 100 000 scenarios; config 5: 4 096 of the 50 000 replicas (a next-event pass over all of them would take minutes; the flow kernel's launch shape -- lists,
ring, FEAT -- does not depend on the replica count).
"""

from __future__ import annotations

import numpy as np
import pytest

from asyncflow_amd import _abi
from asyncflow_amd.plan import lower
from asyncflow_amd.results import differing_scenarios
from oracle import oracle_lib as ol

pytestmark = pytest.mark.gpu


def _sweep(config: int, scenarios: int, extra: list[str]):
    import torch

    import bench

    args = bench.make_parser().parse_args(["--config", str(config), "--scenarios", str(scenarios), *extra])
    args.horizon = args.horizon or None
    wl = bench.build_workload(config, 0, 1, args.scenarios, args.horizon)
    sw = bench.RankSweep(wl, torch.device("cuda", 0), args)
    sw.prepare()
    return sw


def _oracle_picks(sw, k: int) -> None:
    code_name = {v: name for name, v in _abi.PARAM_CODES.items()}
    counts = sw.counts.cpu().numpy().view(np.uint32)
    picks = sorted({int(round(j * (sw.n - 1) / (k - 1))) for j in range(k)})
    for i in picks:
        plan = lower(sw.plan.payload)
        ol.apply_overrides(plan, {(code_name[c], idx): float(col[i]) for c, idx, col, _ in sw.over})
        want = ol.simulate(plan, int(sw.seeds[i]), clock_capacity=sw.clock_cap)
        assert np.array_equal(counts[i, :5].astype(np.uint64), want.counts[:5]), (i, counts[i], want.counts)
        done = int(counts[i, _abi.CNT_COMPLETED])
        got = sw.clock[i, :done].cpu().numpy()
        assert np.array_equal(got.view(np.uint64), want.clock.view(np.uint64)), f"scenario {i}: rqs_clock differs from the oracle"
        ticks = int(counts[i, _abi.CNT_TICKS])
        rows = sw.samples[i, :ticks, : sw.plan.n_series].cpu().numpy().view(np.uint32).T
        assert np.array_equal(rows, want.samples), f"scenario {i}: sampled series differ from the oracle"


@pytest.mark.parametrize(("config", "scenarios"), [(2, 0), (3, 0), (5, 4096)])
def test_every_scenario_of_the_benched_batch_is_identical_on_both_kernel_families(config, scenarios):
    import torch

    flow = _sweep(config, scenarios, [])
    assert flow.n_slices == 1 and not flow.flow_reason
    acc = flow.step()
    torch.cuda.synchronize()
    assert acc["flow_scen"] == flow.n and acc["jit"] >= 1 and acc["jit_fallbacks"] == 0      # the benched launch
    assert flow.n == {2: 10_000, 3: 10_000, 5: 4096}[config]
    c = flow.counts.cpu().numpy().view(np.uint32)
    assert int(np.bitwise_or.reduce(c[:, _abi.CNT_FLAGS])) & _abi.FATAL_FLAGS == 0
    _oracle_picks(flow, 32)

    # the next-event kernels over the same batch, a quarter of the output memory at a time (two full result sets of the
    # grid would be 2 x 80 GB next to the engine's own draw buffers); every slice is compared while it is resident
    seq = _sweep(config, scenarios, ["--no-flow", "--generic-kernels", "--hbm-budget-gb", "24"])
    assert seq.n == flow.n and np.array_equal(seq.seeds, flow.seeds) and seq.clock_cap == flow.clock_cap
    differ = []
    for lo in range(0, seq.n, seq.slice):
        hi = min(seq.n, lo + seq.slice)
        st = seq.run_slice(lo, hi)
        torch.cuda.synchronize()
        assert int(st.flow_scenarios) == 0
        d = differing_scenarios(flow.counts[lo:hi], flow.clock[lo:hi], flow.samples[lo:hi],
                                seq.counts[lo:hi], seq.clock[: hi - lo], seq.samples[: hi - lo])
        differ += (d + lo).tolist()
    assert not differ, f"config {config}: {len(differ)} of {flow.n} scenarios differ between the kernel families (first: {differ[:8]})"
    # the comparison looked at something: completions and ticks of the whole batch
    assert int(c[:, _abi.CNT_COMPLETED].astype(np.int64).sum()) > 1000 * flow.n and int(c[:, _abi.CNT_TICKS].min()) == 11_999
    flow.eng.close()
    seq.eng.close()


def test_every_eighth_scenario_of_config_4_is_identical_on_both_kernel_families():
    """BASELINE config 4 (the users x RTT grid x 10 seeds with event_inj_lb.yml's spikes and outages, 100 000 scenarios) is
    benched in four slices; a next-event pass over all of it takes minutes, so this takes every eighth scenario of the BENCHED
    batch -- same seeds, same columns, every region of the grid, 12 500 scenarios -- through both kernel families and compares
    all of them on the device, plus 16 oracle picks."""
    import torch

    import bench

    def sweep(extra):
        args = bench.make_parser().parse_args(["--config", "4", *extra])
        args.horizon = None
        wl = bench.build_workload(4, 0, 1, 0, None)
        assert wl["n"] == 100_000
        wl["seeds"] = np.ascontiguousarray(wl["seeds"][::8])
        wl["columns"] = {k: np.ascontiguousarray(v[::8]) for k, v in wl["columns"].items()}
        wl["n"] = int(wl["seeds"].size)
        sw = bench.RankSweep(wl, torch.device("cuda", 0), args)
        sw.prepare()
        return sw

    flow = sweep([])
    assert flow.n == 12_500 and flow.n_slices == 1 and not flow.flow_reason
    acc = flow.step()
    torch.cuda.synchronize()
    assert acc["flow_scen"] == flow.n and acc["flow_fallback"][0] == 0
    c = flow.counts.cpu().numpy().view(np.uint32)
    assert int(np.bitwise_or.reduce(c[:, _abi.CNT_FLAGS])) & _abi.FATAL_FLAGS == 0 and int(c[:, _abi.CNT_MARKS].min()) > 0
    _oracle_picks(flow, 16)
    seq = sweep(["--no-flow", "--generic-kernels", "--hbm-budget-gb", "24"])
    differ = []
    for lo in range(0, seq.n, seq.slice):
        hi = min(seq.n, lo + seq.slice)
        st = seq.run_slice(lo, hi)
        torch.cuda.synchronize()
        assert int(st.flow_scenarios) == 0
        d = differing_scenarios(flow.counts[lo:hi], flow.clock[lo:hi], flow.samples[lo:hi],
                                seq.counts[lo:hi], seq.clock[: hi - lo], seq.samples[: hi - lo])
        differ += (d + lo).tolist()
    assert not differ, f"config 4: {len(differ)} of {flow.n} scenarios differ between the kernel families (first: {differ[:8]})"
    flow.eng.close()
    seq.eng.close()


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
