"""Host side of the plan-specialised kernels (asyncflow_amd/jit.py): hipcc cross-compiles gfx950
code objects without a GPU; loading and running them is covered by tests/test_gpu_parity.py."""

from __future__ import annotations

import numpy as np
import pytest

from asyncflow_amd import jit

LB2_SPEC = (
    "-DAF_JIT=1 -DAF_JIT_LDS=1 -DAF_JIT_KLOG=2 -DAF_JIT_METRICS=15 -DAF_JIT_GEN_EDGE=0 -DAF_JIT_CLIENT_EDGE=1 "
    "-DAF_JIT_N_EDGES=6 -DAF_JIT_N_SERVERS=2 -DAF_JIT_LB_ALGO=0 -DAF_JIT_N_LB=2 -DAF_JIT_N_ROWS=6 -DAF_JIT_N_EMARKS=0 "
    "-DAF_JIT_N_SMARKS=0 -DAF_JIT_ORDER_ALL=0 -DAF_JIT_OFF_EDGE=0 -DAF_JIT_OFF_SRV=24 -DAF_JIT_OFF_EP=28 -DAF_JIT_OFF_ROW=32 "
    "-DAF_JIT_OFF_EMARK=50 -DAF_JIT_OFF_SMARK=50 -DAF_JIT_OFF_LB=50 -DAF_JIT_BLOB_BYTES=416 -DAF_JIT_CAP=32 -DAF_JIT_FCAP=16 "
    "-DAF_JIT_OVR_MASK=0 -DAF_JIT_HAS_CLOCK=1 -DAF_JIT_HAS_SAMPLES=1 -DAF_JIT_HAS_ONLINE=0"
)


# what af_engine_jit_spec writes for the BASELINE config-2 sweep on the stage-parallel kernel (af_flow_kernel<1, 0>, 64-entry
# lists, a 32-row LDS tick ring: 9 512 B of LDS per wave)
LB2_FLOW_SPEC = (
    "-DAF_JIT=1 -DAF_FLOW_JIT=1 -DAF_FJ_IPL=1 -DAF_FJ_FEAT=0 -DAF_FJ_TOTAL_TIME=0x4082c00000000000ull "
    "-DAF_FJ_PERIOD=0x3fa999999999999aull -DAF_FJ_INV_PERIOD=0x4034000000000000ull "
    "-DAF_FJ_TICK_EPS=0x3e112e0be826d695ull -DAF_FJ_METRICS=15 -DAF_FJ_GEN_EDGE=0 -DAF_FJ_CLIENT_EDGE=1 "
    "-DAF_FJ_N_EDGES=6 -DAF_FJ_N_SERVERS=2 -DAF_FJ_HAS_LB=1 -DAF_FJ_N_LB=2 -DAF_FJ_N_EMARKS=0 "
    "-DAF_FJ_N_SMARKS=0 -DAF_FJ_LC=0 -DAF_FJ_MAX_PRE=0 -DAF_FJ_MAX_CPU=1 -DAF_FJ_MAX_POST=1 -DAF_FJ_OFF_EDGE=0 "
    "-DAF_FJ_OFF_SRV=24 -DAF_FJ_OFF_EP=28 -DAF_FJ_OFF_ROW=32 -DAF_FJ_OFF_EMARK=50 -DAF_FJ_OFF_SMARK=50 "
    "-DAF_FJ_OFF_LB=50 -DAF_FJ_BLOB_BYTES=416 -DAF_FJ_N_TICKS=11999 -DAF_FJ_HAS_CLOCK=1 -DAF_FJ_HAS_SAMPLES=1 "
    "-DAF_FJ_HAS_ONLINE=0 -DAF_FJ_HAS_OVR=0 -DAF_FJ_DIST_ALL=255 -DAF_FJ_RAM_SCALE=1.0 "
    "-DAF_FJ_LAYOUT=64,32,16,16,1,12,2,0,0,512,528,544,704,768,544,864,866,898,965,1157,64,64,64,64,0,128,256,384,840,0"
)


def test_specialised_flow_kernel_builds_with_one_entry_point(tmp_path, monkeypatch):
    monkeypatch.setattr(jit, "CACHE_DIR", tmp_path)
    image = jit.code_object(LB2_FLOW_SPEC)
    assert image.startswith(b"__CLANG_OFFLOAD_BUNDLE__") and b"gfx950" in image[:4096]
    assert b"af_flow_jit" in image and b"af_jit_lean" not in image
    with pytest.raises(jit.JitUnavailableError, match="not in the cache"):
        jit.code_object(LB2_FLOW_SPEC.replace("-DAF_FJ_IPL=1", "-DAF_FJ_IPL=2"), build=False)
    assert jit.code_object(LB2_FLOW_SPEC, build=False) == image     # cache hit without a build


def test_specialised_code_object_builds_caches_and_exports_the_three_entry_points(tmp_path, monkeypatch):
    monkeypatch.setattr(jit, "CACHE_DIR", tmp_path)
    image = jit.code_object(LB2_SPEC)
    assert image.startswith(b"__CLANG_OFFLOAD_BUNDLE__") and b"gfx950" in image[:4096]
    for name in (b"af_jit_lean", b"af_jit_order3", b"af_jit_order2"):
        assert name in image
    cached = list(tmp_path.glob("*.hsaco"))
    assert len(cached) == 1 and not list(tmp_path.glob("*.tmp"))
    assert jit.code_object(LB2_SPEC) == image                    # second call: from the cache
    assert len(list(tmp_path.glob("*.hsaco"))) == 1
    other = jit.code_object(LB2_SPEC.replace("-DAF_JIT_KLOG=2", "-DAF_JIT_KLOG=0"))
    assert other != image and len(list(tmp_path.glob("*.hsaco"))) == 2


def test_a_short_sweep_leaves_the_build_to_a_background_thread(tmp_path, monkeypatch):
    """A sweep too short to repay ~4 s of hipcc runs on the generic kernels and starts the build beside it: the next sweep of the
    same shape finds the specialised kernel in the cache (round 5; the generic form of BASELINE config 2 costs 64 instead of 38 ms)."""
    monkeypatch.setattr(jit, "CACHE_DIR", tmp_path)
    monkeypatch.setattr(jit, "_background", {})
    monkeypatch.setenv("ASYNCFLOW_JIT_BACKGROUND", "1")
    with pytest.raises(jit.JitUnavailableError, match="not in the cache"):
        jit.code_object(LB2_FLOW_SPEC, build=False)
    t = jit.build_in_background(LB2_FLOW_SPEC)
    assert t is not None and jit.build_in_background(LB2_FLOW_SPEC) is t          # once per spec
    t.join(timeout=120)
    assert not t.is_alive()
    assert jit.code_object(LB2_FLOW_SPEC, build=False).startswith(b"__CLANG_OFFLOAD_BUNDLE__")
    monkeypatch.setenv("ASYNCFLOW_JIT_BACKGROUND", "0")
    assert jit.build_in_background(LB2_FLOW_SPEC.replace("-DAF_FJ_IPL=1", "-DAF_FJ_IPL=2")) is None


def test_a_failing_build_is_reported_as_unavailable_not_as_a_crash(tmp_path, monkeypatch):
    monkeypatch.setattr(jit, "CACHE_DIR", tmp_path)
    with pytest.raises(jit.JitUnavailableError, match="hipcc --genco failed"):
        jit.code_object("-DAF_JIT=1 -DAF_JIT_LDS=1")            # constants missing: does not compile
    assert not list(tmp_path.iterdir())


def test_an_unwritable_cache_is_reported_as_unavailable(tmp_path, monkeypatch):
    blocker = tmp_path / "file"
    blocker.write_text("x")                                       # a FILE where the cache directories should be
    monkeypatch.setattr(jit, "CACHE_DIR", blocker / "a")
    monkeypatch.setattr(jit, "_FALLBACK_CACHE_DIR", blocker / "b")
    with pytest.raises(jit.JitUnavailableError, match="no writable cache"):
        jit.code_object(LB2_SPEC)


def test_the_runner_never_asks_for_a_fifo_the_engine_refuses():
    """saturated payloads used to estimate fifo = 65535 -> AF_ERR_CAPACITY at engine creation."""
    from asyncflow_amd import _abi
    from asyncflow_amd.runner import SimulationRunner, _fifo_pow2
    from oracle.scenarios import overload

    r = SimulationRunner(simulation_input=overload(horizon=600))
    cap, fifo, _ = r._capacities([])  # noqa: SLF001
    assert cap <= _abi.MAX_REQUEST_CAPACITY and fifo <= _abi.MAX_FIFO_CAPACITY and fifo & (fifo - 1) == 0
    assert _fifo_pow2(10**9) == _abi.MAX_FIFO_CAPACITY and _fifo_pow2(9) == 16


# ---- planning-only engine (AF_DEVICE_PLAN_ONLY): the spec without a device -------------------------------------------
def _lb2_spec_on(device: int, n: int = 10_000) -> str:
    import numpy as np

    from asyncflow_amd.engine import Engine
    from asyncflow_amd.plan import lower
    from oracle.scenarios import lb_two_servers

    plan = lower(lb_two_servers())
    eng = Engine(plan, device)
    try:
        return eng.jit_spec(0x5EED0000 + np.arange(n, dtype=np.uint64), [], clock_ptr=8, clock_capacity=plan.clock_capacity(None),
                            samples_ptr=8, tick_capacity=plan.tick_count, counts_ptr=8, draw_capacity=plan.clock_capacity(None))
    finally:
        eng.close()


def test_a_planning_only_engine_writes_the_flow_spec_without_a_device():
    from asyncflow_amd.engine import PLAN_ONLY

    spec = _lb2_spec_on(PLAN_ONLY)
    assert spec.startswith("-DAF_JIT=1 -DAF_FLOW_JIT=1 -DAF_FJ_IPL=1 -DAF_FJ_FEAT=0 ") and "-DAF_FJ_N_TICKS=11999 " in spec
    assert spec == _lb2_spec_on(PLAN_ONLY, n=64)            # (replica sweeps: the scenario count is a launch argument)


def test_a_planning_only_engine_cannot_run():
    import numpy as np

    from asyncflow_amd.engine import PLAN_ONLY, Engine, EngineUnavailableError
    from asyncflow_amd.plan import lower
    from oracle.scenarios import lb_two_servers

    eng = Engine(lower(lb_two_servers(horizon=5)), PLAN_ONLY)
    counts = np.zeros((1, 8), dtype=np.uint32)
    with pytest.raises(EngineUnavailableError, match="planning-only"):
        eng.run(np.array([1], dtype=np.uint64), [], clock_ptr=0, clock_capacity=0, samples_ptr=0, tick_capacity=0,
                counts_ptr=counts.ctypes.data)
    assert eng.flow_reason() == ""
    eng.close()


def test_a_runner_names_and_prebuilds_its_kernel_without_a_gpu(monkeypatch):
    """SimulationRunner.jit_spec() / prebuild(): the spec of the sweep's first run from a planning-only engine -- for BASELINE
    config 2 the very spec bench.prebuild_kernels() builds -- handed to the compiler cache; sweep columns make their own kernel."""
    import numpy as np

    import bench
    from asyncflow_amd.runner import SimulationRunner
    from asyncflow_amd.workloads import lb_two_servers

    seeds = np.arange(64, dtype=np.uint64) + 0x5EED0000
    runner = SimulationRunner(simulation_input=lb_two_servers(horizon=600), seeds=seeds)
    spec = runner.jit_spec()
    built = []
    monkeypatch.setattr(jit, "code_object", lambda sp, build=True: built.append(sp) or b"")
    assert bench.prebuild_kernels(configs=(2,), worlds=(1,), verbose=False) == [spec]
    assert runner.prebuild() == spec and built == [spec, spec]
    swept = SimulationRunner(simulation_input=lb_two_servers(horizon=600), seeds=seeds,
                             sweep={"rqs_input.avg_active_users.mean": np.linspace(100.0, 700.0, 64)})
    assert "-DAF_FJ_HAS_OVR=1" in swept.jit_spec() and "-DAF_FJ_HAS_OVR=0" in spec
    from asyncflow_amd.engine import EngineUnavailableError

    with pytest.raises(EngineUnavailableError, match="only sweeps of the stage-parallel kernel"):   # (their shape depends on the device)
        SimulationRunner(simulation_input=lb_two_servers(horizon=600), seeds=seeds, flow=False).jit_spec()


def test_prebuild_fills_the_cache_for_every_default_bench_line(tmp_path, monkeypatch):
    """`__graft_entry__.build()` calls bench.prebuild_kernels(): afterwards a box without hipcc still finds the kernels."""
    import bench

    monkeypatch.setattr(jit, "CACHE_DIR", tmp_path)
    monkeypatch.setattr(jit, "_FALLBACK_CACHE_DIR", tmp_path / "none")
    specs = bench.prebuild_kernels(configs=(2,), worlds=(1,), verbose=False)
    assert len(specs) == 1 and len(list(tmp_path.glob("*.hsaco"))) == 1
    monkeypatch.setenv("ASYNCFLOW_NO_HIPCC", "1")
    assert jit.code_object(specs[0]).startswith(b"__CLANG_OFFLOAD_BUNDLE__")     # a cache hit needs no compiler
    with pytest.raises(jit.JitUnavailableError, match="ASYNCFLOW_NO_HIPCC"):
        jit.code_object(specs[0].replace("-DAF_FJ_IPL=1", "-DAF_FJ_IPL=2"))


@pytest.mark.parametrize("config", [2, 6])
def test_the_bench_kernels_keep_their_state_out_of_scratch_memory(config, tmp_path):
    """A run-time index into a member array of the kernel's argument / state objects makes the compiler address the whole object
    indirectly, i.e. keep it in scratch memory (round 5 met it twice: the generic server tiers at 437 instead of 102 ms, and the
    general-server workload at 441 instead of 279 ms when the client's target list became a run-time value).  The plan-specialised
    kernels of the bench's workloads must stay within a few call frames' worth of scratch: `-Rpass-analysis=kernel-resource-usage`."""
    import re
    import shlex
    import subprocess

    import numpy as np

    import bench
    from asyncflow_amd.build import CSRC, hipcc_path
    from asyncflow_amd.engine import PLAN_ONLY, Engine

    args = bench.make_parser().parse_args(["--config", str(config)])
    args.horizon = None
    shape = bench.rank_shape(bench.build_workload(config, 0, 1, 0, None), args)
    eng = Engine(shape["plan"], PLAN_ONLY, **shape["engine_kw"])
    over = [(c, i, np.ascontiguousarray(v[: shape["slice"]])) for c, i, v, _ in shape["over"]]
    spec = eng.jit_spec(shape["seeds"][: shape["slice"]], over, clock_ptr=8, clock_capacity=shape["clock_cap"], samples_ptr=8,
                        tick_capacity=shape["ticks"], counts_ptr=8, draw_capacity=shape["clock_cap"])
    eng.close()
    cmd = [hipcc_path(), *jit._FLAGS, *shlex.split(spec), "-Rpass-analysis=kernel-resource-usage", "-o", str(tmp_path / "k.hsaco"),  # noqa: SLF001
           str(CSRC / "engine.hip")]
    res = subprocess.run(cmd, capture_output=True, text=True, check=False)
    assert res.returncode == 0, res.stderr[-2000:]
    block = res.stderr[res.stderr.index("Function Name: af_flow_jit"):]
    scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", block).group(1))
    assert scratch <= 256, f"af_flow_jit of config {config} keeps {scratch} B per lane in scratch memory"


def test_key_schedule_per_call_is_asked_for_where_it_was_measured_to_pay():
    """Round 5: Philox's key schedule on the scalar unit at every call site (`-DAF_FJ_KEYS_PER_CALL`, af_flow.hpp: Flow::edge_draw)
    is part of the spec of the lean form and of the forms with timeline marks -- BASELINE configs 2 and 4, where it is faster -- and
    of no other (configs 3, 5 and the general-server workload: profiles/r05/philox_keys_ab.txt)."""
    import bench

    from asyncflow_amd.engine import PLAN_ONLY, Engine

    asked = {}
    for cfg in (2, 3, 4, 5, 6):
        args = bench.make_parser().parse_args(["--config", str(cfg)])
        args.horizon = None
        shape = bench.rank_shape(bench.build_workload(cfg, 0, 1, 0, None), args)
        eng = Engine(shape["plan"], PLAN_ONLY, **shape["engine_kw"])
        hi = min(shape["slice"], shape["n"])
        over = [(c, i, np.ascontiguousarray(v[:hi])) for c, i, v, _ in shape["over"]]
        spec = eng.jit_spec(shape["seeds"][:hi], over, clock_ptr=8, clock_capacity=shape["clock_cap"], samples_ptr=8,
                            tick_capacity=shape["ticks"], counts_ptr=8, draw_capacity=shape["clock_cap"])
        eng.close()
        asked[cfg] = "-DAF_FJ_KEYS_PER_CALL=1" in spec
    assert asked == {2: True, 3: False, 4: True, 5: False, 6: False}


def test_second_chance_tier_form_builds_with_the_key_schedule_per_call(tmp_path, monkeypatch):
    """The first form of that change (an `asm` constraint on the seed without a `v_readfirstlane` in front) crashed the compiler
    on a FEAT_CHAIN instantiation with timeline marks (the GPU suite's specialised-build test found it): the build of such a
    plan is part of the CPU suite now."""
    import random

    from asyncflow_amd.runner import SimulationRunner
    from oracle.scenarios import server_tiers

    monkeypatch.setattr(jit, "CACHE_DIR", tmp_path)
    runner = SimulationRunner(simulation_input=server_tiers(random.Random(91001), horizon=12), seeds=np.arange(8, dtype=np.uint64) + 50,
                              specialise=True, on_negative_delay="flag")
    spec = runner.prebuild()
    assert "-DAF_FJ_KEYS_PER_CALL=1" in spec and "-DAF_FJ_FEAT=607" in spec
    assert len(list(tmp_path.glob("*.hsaco"))) == 1


def test_the_references_own_examples_find_their_kernel_in_the_shipped_cache():
    """a user's FIRST sweep ran on the generic kernels (64 instead of 38 ms per 10 000 replicas of
    two_servers_lb.yml) unless it was long enough to repay a hipcc run.  The kernel a sweep launches does not depend on its
    replica count, so `__graft_entry__.build()` now builds ONE kernel per example input of the reference
    (`SimulationRunner.prebuild_reference_examples`): whoever runs the reference's examples -- the first thing a user of the
    reference does -- loads a specialised kernel from the cache that ships with the package, with no compiler in reach."""
    from asyncflow_amd import jit, workloads
    from asyncflow_amd.runner import SimulationRunner

    specs = SimulationRunner.prebuild_reference_examples()
    assert set(specs) == set(workloads.reference_examples()) and len(specs) == 6
    for name, payload in workloads.reference_examples().items():
        for replicas in (1, 250, 10_000):
            spec = SimulationRunner(simulation_input=payload, replicas=replicas).jit_spec()
            assert spec == specs[name], (name, replicas)                 # one shape, whatever the number of replicas
        assert "-DAF_FLOW_JIT=1" in specs[name]                          # every example is in the stage-parallel kernel's range
        assert len(jit.code_object(specs[name], build=False)) > 10_000    # ... and its code object is in the cache
