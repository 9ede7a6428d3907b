"""Build + load the TEST-ONLY host instantiation of the engine core (see hostcheck.cpp)."""

from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

from asyncflow_amd import _abi
from asyncflow_amd.plan import DevicePlan

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
LIB = HERE / "libaf_hostcheck.so"
_lib = None


def build(force: bool = False) -> Path:
    srcs = [HERE / "hostcheck.cpp", HERE / "wave_emul.hpp", ROOT / "asyncflow_amd/csrc/af_core.hpp",
            ROOT / "asyncflow_amd/csrc/af_math.hpp", ROOT / "asyncflow_amd/csrc/af_plan_pack.hpp",
            ROOT / "asyncflow_amd/csrc/af_flow.hpp", ROOT / "asyncflow_amd/csrc/af_flow_host.hpp",
            ROOT / "asyncflow_amd/csrc/af_pregen.hpp",
            ROOT / "include/asyncflow_hip.h"]
    newest = max(p.stat().st_mtime for p in srcs)
    if force or not LIB.exists() or LIB.stat().st_mtime < newest:
        subprocess.run(
            ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
             "-Wall", "-o", str(LIB), str(srcs[0])],
            check=True, capture_output=True, text=True,
        )
    return LIB


def lib() -> C.CDLL:
    global _lib  # noqa: PLW0603
    if _lib is None:
        build()
        L = C.CDLL(str(LIB))
        L.hc_simulate.argtypes = [
            C.POINTER(_abi.AfPlan), C.c_uint64, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
            C.POINTER(C.c_double), C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_double), C.c_uint32,
            C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_uint32,
        ]
        L.hc_simulate.restype = C.c_int
        L.hc_set_online.argtypes = [C.POINTER(C.c_uint32), C.c_uint32, C.c_double, C.POINTER(C.c_uint32), C.c_uint32]
        L.hc_set_online.restype = None
        L.hc_set_two_pass.argtypes = [C.c_int]
        L.hc_set_two_pass.restype = None
        L.hc_reruns.argtypes = []
        L.hc_reruns.restype = C.c_int
        L.hc_bytes_per_lane.argtypes = [C.c_uint32] * 7
        L.hc_bytes_per_lane.restype = C.c_uint64
        L.hc_flow_simulate.argtypes = [
            C.POINTER(_abi.AfPlan), C.c_uint64, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
            C.POINTER(C.c_double), C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_double), C.c_uint32,
            C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_uint32,
        ]
        L.hc_flow_simulate.restype = C.c_int
        L.hc_set_test_quantum.argtypes = [C.c_int]
        L.hc_set_test_quantum.restype = None
        L.hc_flow_reason.argtypes = []
        L.hc_flow_reason.restype = C.c_char_p
        L.hc_flow_lds_bytes.argtypes = [C.POINTER(_abi.AfPlan), C.c_uint32, C.c_uint32]
        L.hc_flow_lds_bytes.restype = C.c_uint64
        L.hc_arrivals.argtypes = [C.c_int, C.c_uint64, C.c_uint32] + [C.c_double] * 5 + [C.c_uint32, C.POINTER(C.c_double),
                                  C.POINTER(C.c_uint32)]
        L.hc_arrivals.restype = C.c_int64
        L.hc_math.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_uint64]
        L.hc_math.restype = None
        _lib = L
    return _lib


def arrivals(which: int, seed: int, *, dist: int, mean: float, sigma: float, rpm: float, window_s: float, horizon: float,
             n_draw: int) -> tuple[int, np.ndarray, int]:
    """The arrival sampler on the host: 0 = af::gen_next_gap (sequential statement), 1 / 2 = the per-lane functions of
    af_arrival_groups (af_pregen.hpp) with the hoisted-reciprocal / plain division.  -> (arrivals, times[n_draw], flags)"""
    out = np.empty(n_draw, dtype=np.float64)
    flags = C.c_uint32(0)
    n = lib().hc_arrivals(which, C.c_uint64(seed), dist, mean, sigma, rpm, window_s, horizon, n_draw,
                          out.ctypes.data_as(C.POINTER(C.c_double)), C.byref(flags))
    return int(n), out, int(flags.value)


def simulate(plan: DevicePlan, seed: int, *, cap: int = 4096, fcap: int = 4096,
             overrides: list[tuple[str, int, float]] | None = None, clock_capacity: int | None = None,
             draw_capacity: int | None = None):
    """Run one scenario through the engine core on the host. Returns (counts, clock, samples)."""
    L = lib()
    cplan = plan.as_ctypes()
    ov = overrides or []
    params = np.asarray([_abi.PARAM_CODES[o[0]] for o in ov], dtype=np.uint32)
    idxs = np.asarray([o[1] for o in ov], dtype=np.uint32)
    vals = np.asarray([o[2] for o in ov], dtype=np.float64)
    ccap = int(clock_capacity if clock_capacity is not None else plan.clock_capacity())
    clock = np.zeros((ccap, 2), dtype=np.float64)
    ticks = max(plan.tick_count, 1)
    samples = np.zeros((ticks, plan.series_pitch), dtype=np.uint32)  # device layout [tick][pitch]
    counts = np.zeros(_abi.CNT_SLOTS, dtype=np.uint32)
    u32p, f64p = C.POINTER(C.c_uint32), C.POINTER(C.c_double)
    rc = L.hc_simulate(
        C.byref(cplan), C.c_uint64(seed), len(ov), params.ctypes.data_as(u32p), idxs.ctypes.data_as(u32p),
        vals.ctypes.data_as(f64p), cap, fcap, ccap, clock.ctypes.data_as(f64p), ticks,
        samples.ctypes.data_as(u32p), counts.ctypes.data_as(u32p),
        int(draw_capacity if draw_capacity is not None else plan.clock_capacity()),
    )
    if rc != 0:
        msg = f"hc_simulate failed: {rc}"
        raise RuntimeError(msg)
    n = int(counts[_abi.CNT_COMPLETED])
    k = int(counts[_abi.CNT_TICKS])
    return counts, clock[: min(n, ccap)].copy(), np.ascontiguousarray(samples[:k, : plan.n_series].T)


FLOW_FALLBACK = 1 << 8
FLOW_WHY = {1 << 9: "tie", 1 << 10: "list", 1 << 11: "ring", 1 << 12: "ram"}


def flow_simulate(plan: DevicePlan, seed: int, *, ipl: int = 1, ring_rows: int = 64, robust: bool = False, far: bool = True,
                  long_list_entries: int = 256, long_list: int | None = None,
                  overrides: list[tuple[str, int, float]] | None = None, clock_capacity: int | None = None,
                  draw_capacity: int | None = None):
    """Run one scenario through the stage-parallel kernel (af_flow.hpp) on the 64-fibre wave emulator.

    Returns (counts, clock, samples) like :func:`simulate`, or ``None`` when the plan is not eligible
    (``flow_reason()`` says why).  ``robust``: the second-chance instantiation (lists that carry the send times:
    equal delivery times at a station are ordered like SimPy orders them; ``long_list_entries`` for station list
    ``long_list`` -- or all four --, 256 for the others).  ``far=False``: the lean instantiations without FEAT_FAR (the sender
    enters both ends of every message; a delivery beyond the tick ring hands the scenario back).
    ``counts[CNT_FLAGS] & FLOW_FALLBACK``: the kernel handed the scenario
    back to the sequential kernels (outputs are then incomplete)."""
    L = lib()
    cplan = plan.as_ctypes()
    ov = overrides or []
    params = np.asarray([_abi.PARAM_CODES[o[0]] for o in ov], dtype=np.uint32)
    idxs = np.asarray([o[1] for o in ov], dtype=np.uint32)
    vals = np.asarray([o[2] for o in ov], dtype=np.float64)
    ccap = int(clock_capacity if clock_capacity is not None else plan.clock_capacity())
    clock = np.zeros((ccap, 2), dtype=np.float64)
    ticks = max(plan.tick_count, 1)
    samples = np.zeros((ticks, plan.series_pitch), dtype=np.uint32)
    counts = np.zeros(_abi.CNT_SLOTS, dtype=np.uint32)
    u32p, f64p = C.POINTER(C.c_uint32), C.POINTER(C.c_double)
    rc = L.hc_flow_simulate(
        C.byref(cplan), C.c_uint64(seed), len(ov), params.ctypes.data_as(u32p), idxs.ctypes.data_as(u32p),
        vals.ctypes.data_as(f64p),
        ipl | (0 if far else 0x200) | ((0x100 | (long_list_entries << 16) | ((0 if long_list is None else long_list + 1) << 12)) if robust else 0), ring_rows, ccap, clock.ctypes.data_as(f64p), ticks,
        samples.ctypes.data_as(u32p), counts.ctypes.data_as(u32p),
        int(draw_capacity if draw_capacity is not None else plan.clock_capacity()),
    )
    if rc == 1:
        return None
    if rc != 0:
        msg = f"hc_flow_simulate failed: {rc}"
        raise RuntimeError(msg)
    n = int(counts[_abi.CNT_COMPLETED])
    k = int(counts[_abi.CNT_TICKS])
    return counts, clock[: min(n, ccap)].copy(), np.ascontiguousarray(samples[:k, : plan.n_series].T)


def flow_reason() -> str:
    return (lib().hc_flow_reason() or b"").decode()


def set_test_quantum(bits: int) -> None:
    """TEST-ONLY tie generator of the host builds (asyncflow_amd/csrc/af_math.hpp::test_quant)."""
    lib().hc_set_test_quantum(int(bits))


def math(kind: int, x: np.ndarray) -> np.ndarray:
    """af_math.hpp's elementary functions as the host compiles them (1 log, 2 exp, 3 normal quantile, 7 log on (0, 1])."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    lib().hc_math(kind, x.ctypes.data_as(C.POINTER(C.c_double)), out.ctypes.data_as(C.POINTER(C.c_double)), x.size)
    return out
