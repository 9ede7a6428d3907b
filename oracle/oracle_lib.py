"""ctypes front-end of oracle/libaf_oracle.so -- ORACLE / TEST INFRASTRUCTURE ONLY.

Only tests/, ``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline`` leg may
import this module.  ``asyncflow_amd`` never does.
"""

from __future__ import annotations

import ctypes as C
import subprocess
from dataclasses import dataclass
from pathlib import Path

import numpy as np

from asyncflow_amd import _abi
from asyncflow_amd.plan import DevicePlan

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "libaf_oracle.so"
_lib: C.CDLL | None = None


def build(force: bool = False) -> Path:
    """Compile the C restatement (gcc, seconds)."""
    src_m = max([(_HERE / n).stat().st_mtime for n in ("des_oracle.c", "oracle_rng.h")]
                + [(_HERE.parent / "include" / "asyncflow_hip.h").stat().st_mtime])
    if force or not _LIB_PATH.exists() or _LIB_PATH.stat().st_mtime < src_m:
        subprocess.run(["make", "-C", str(_HERE), "-B", "libaf_oracle.so"], check=True, capture_output=True)
    return _LIB_PATH


def lib() -> C.CDLL:
    global _lib  # noqa: PLW0603
    if _lib is None:
        build()
        L = C.CDLL(str(_LIB_PATH))
        L.orc_simulate.argtypes = [
            C.POINTER(_abi.AfPlan), C.c_uint64, C.c_uint64, C.POINTER(C.c_double),
            C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64),
        ]
        L.orc_simulate.restype = C.c_int
        L.orc_x_uniform.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_x_uniform.restype = C.c_double
        L.orc_x_word0.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
        L.orc_x_word0.restype = C.c_uint32
        L.orc_x_philox.argtypes = [C.c_uint32] * 6 + [C.POINTER(C.c_uint32)]
        L.orc_x_philox.restype = None
        for name in ("orc_x_log", "orc_x_exp", "orc_x_norminv"):
            getattr(L, name).argtypes = [C.c_double]
            getattr(L, name).restype = C.c_double
        L.orc_x_poisson.argtypes = [C.c_double, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_x_poisson.restype = C.c_int64
        L.orc_x_variate.argtypes = [C.c_int, C.c_double, C.c_double, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_x_variate.restype = C.c_double
        L.orc_set_test_quantum.argtypes = [C.c_int]
        L.orc_set_test_quantum.restype = None
        L.orc_last_ties.argtypes = []
        L.orc_last_ties.restype = C.c_uint64
        L.orc_last_heap_events.argtypes = []
        L.orc_last_heap_events.restype = C.c_uint64
        L.orc_last_put_waits.argtypes = []
        L.orc_last_put_waits.restype = C.c_uint64
        L.orc_tick_count.argtypes = [C.c_double, C.c_double]
        L.orc_tick_count.restype = C.c_uint32
        _lib = L
    return _lib


STREAM_GENERATOR = 0


def stream_edge(e: int) -> int:
    return 1 + e


def stream_server(s: int) -> int:
    return 0x1000 + s


@dataclass
class OracleResult:
    """One scenario simulated by the C oracle."""

    counts: np.ndarray   # uint64[8], af_count_slot order
    clock: np.ndarray    # float64[n_completed, 2]  (start, finish)
    samples: np.ndarray  # uint32[n_series, n_ticks] raw words (ram rows are float32 bits)
    ties: int = 0        # timed events popped at the timestamp of the previous one
    heap_events: int = 0 # SimPy-equivalent heap pushes
    put_waits: int = 0   # RAM puts that simpy's Container refused at first (fractional needs; des_oracle.c::ram_trigger_put)

    @property
    def generated(self) -> int:
        return int(self.counts[_abi.CNT_GENERATED])

    @property
    def completed(self) -> int:
        return int(self.counts[_abi.CNT_COMPLETED])

    @property
    def dropped(self) -> int:
        return int(self.counts[_abi.CNT_DROPPED])

    @property
    def events(self) -> int:
        return int(self.counts[_abi.CNT_EVENTS])

    @property
    def ticks(self) -> int:
        return int(self.counts[_abi.CNT_TICKS])


def apply_overrides(plan: DevicePlan, overrides: dict[tuple[str, int], float]) -> None:
    """Write per-scenario parameter values into the plan arrays (in place)."""
    for (name, index), value in overrides.items():
        if name == "gen_users_mean":
            plan.gen_users_mean = float(value)
        elif name == "gen_users_sigma":
            plan.gen_users_sigma = float(value)
        elif name == "gen_rpm_mean":
            plan.gen_rpm_mean = float(value)
        elif name == "edge_mean":
            plan.edge_mean[index] = value
        elif name == "edge_sigma":
            plan.edge_sigma[index] = value
        elif name == "edge_dropout":
            plan.edge_dropout[index] = value
        elif name == "step_time":
            plan.step_time[index] = value
        elif name == "gen_window":
            plan.gen_window_s = float(value)
        elif name == "srv_cores":
            plan.srv_cores[index] = int(value)
        elif name == "srv_ram_mb":
            plan.srv_ram_mb[index] = value
        elif name in ("emark_time", "emark_delta", "smark_time"):
            getattr(plan, name)[index] = value
        elif name == "emark_edge":
            plan.emark_edge[index] = int(value)
        elif name == "smark_lb_edge":
            plan.smark_lb_edge[index] = int(value)
        elif name == "smark_down":
            plan.smark_down[index] = int(value != 0)
        else:
            msg = f"unknown override {name!r}"
            raise KeyError(msg)


def simulate(
    plan: DevicePlan,
    seed: int,
    *,
    clock_capacity: int | None = None,
    want_samples: bool = True,
    want_clock: bool = True,
) -> OracleResult:
    """Run ONE scenario of ``plan`` with Philox key ``seed`` on the CPU: the SimPy-faithful
    restatement (des_oracle.c) == the reference, exact timestamp ties included (``ties`` counts
    the timed events that shared their timestamp with the previous one)."""
    L = lib()
    cplan = plan.as_ctypes()
    cap = int(clock_capacity if clock_capacity is not None else plan.clock_capacity())
    clock = np.zeros((cap, 2), dtype=np.float64) if want_clock else None
    ticks = plan.tick_count
    samples = np.zeros((plan.n_series, max(ticks, 1)), dtype=np.uint32) if want_samples else None
    counts = np.zeros(_abi.CNT_SLOTS, dtype=np.uint64)
    rc = L.orc_simulate(
        C.byref(cplan),
        C.c_uint64(seed),
        C.c_uint64(cap),
        clock.ctypes.data_as(C.POINTER(C.c_double)) if clock is not None else None,
        C.c_uint64(max(ticks, 1)),
        samples.ctypes.data_as(C.POINTER(C.c_uint32)) if samples is not None else None,
        counts.ctypes.data_as(C.POINTER(C.c_uint64)),
    )
    if rc != 0:
        msg = f"orc_simulate failed with {rc}"
        raise RuntimeError(msg)
    n = int(counts[_abi.CNT_COMPLETED])
    return OracleResult(
        counts=counts,
        clock=clock[: min(n, cap)].copy() if clock is not None else np.zeros((0, 2)),
        samples=samples[:, : int(counts[_abi.CNT_TICKS])].copy() if samples is not None else np.zeros((0, 0), np.uint32),
        ties=int(L.orc_last_ties()),
        heap_events=int(L.orc_last_heap_events()),
        put_waits=int(L.orc_last_put_waits()),
    )


def set_test_quantum(bits: int) -> None:
    """TEST-ONLY: round every latency / arrival gap down to a multiple of 2**-bits s (0 = off) -- exact
    timestamp ties by the thousand; tests/hostcheck has the same switch (build.set_test_quantum)."""
    lib().orc_set_test_quantum(int(bits))
