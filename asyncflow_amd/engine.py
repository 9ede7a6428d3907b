"""ctypes binding of the HIP engine (include/asyncflow_hip.h).

There is NO CPU fallback: if the shared library is missing or no MI355X is
visible, :class:`EngineUnavailableError` is raised.
"""

from __future__ import annotations

import ctypes as C
import os
from pathlib import Path
from typing import Sequence

import numpy as np

from . import _abi
from . import build as af_build
from .build import LIB_PATH
from .plan import DevicePlan


class EngineUnavailableError(RuntimeError):
    """The HIP library or the GPU is missing (the engine never falls back to the CPU)."""


class EngineError(RuntimeError):
    """A C-ABI call failed; the message is ``af_last_error()``."""


_lib: C.CDLL | None = None


def _load_hip_runtime() -> None:
    """Make ONE HIP runtime globally visible before dlopen()ing the engine.

    libasyncflow_hip.so is linked with --no-hip-rt: its hip* symbols bind to the
    runtime already in the process.  Under PyTorch that must be the copy torch
    bundles (torch/lib/libamdhip64.so), otherwise torch tensors and the engine
    would live in two different HIP/HSA runtimes.  Without torch, the system ROCm
    runtime is used; a C/C++ host links its own.
    """
    candidates: list[str] = []
    try:
        import torch

        candidates.append(str(Path(torch.__file__).resolve().parent / "lib" / "libamdhip64.so"))
    except Exception:  # noqa: BLE001 - torch is plumbing, not a requirement of the C ABI
        pass
    candidates += ["libamdhip64.so.7", "/opt/rocm/lib/libamdhip64.so", "libamdhip64.so"]
    for cand in candidates:
        if cand.startswith("/") and not Path(cand).exists():
            continue
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            continue
        else:
            return
    msg = "no HIP runtime (libamdhip64) could be loaded"
    raise EngineUnavailableError(msg)


def load_library(path: str | Path | None = None) -> C.CDLL:
    global _lib  # noqa: PLW0603
    if _lib is None:
        override = path or os.environ.get("ASYNCFLOW_HIP_LIB")            # env override: A/B builds in experiments
        p = Path(override or LIB_PATH)
        if not override and af_build.needs_build():
            # Build on demand: the library is git-ignored, so a fresh checkout (or a tree whose sources changed since the
            # last build -- content hash, not mtime) has none.  Without hipcc a stale library is an error as well: running
            # other code than the sources say is worse than not running.
            if not af_build.have_hipcc():
                if p.exists() and not af_build.STAMP_PATH.exists():
                    # a library shipped prebuilt (or built by a tree that wrote no stamp): nothing says it is stale, and nothing
                    # here could rebuild it -- load it; the ABI version is checked below either way
                    import warnings

                    warnings.warn(f"{p} has no build stamp and hipcc is not available: loading it as it is", RuntimeWarning, stacklevel=2)
                else:
                    what = "not found" if not p.exists() else "was built from other sources than the tree holds"
                    msg = (f"{p} {what} and hipcc is not available to rebuild it (`python -m asyncflow_amd.build`, "
                           "hipcc --offload-arch=gfx950). asyncflow_amd has no CPU fallback.")
                    raise EngineUnavailableError(msg)
            else:
                try:
                    af_build.build()
                except (RuntimeError, OSError) as exc:      # hipcc failed / a read-only tree
                    msg = f"building {p} failed: {type(exc).__name__}: {exc}"
                    raise EngineUnavailableError(msg) from exc
        if not p.exists():
            msg = f"{p} not found. asyncflow_amd has no CPU fallback."
            raise EngineUnavailableError(msg)
        _load_hip_runtime()
        try:
            lib = C.CDLL(str(p))
        except OSError as exc:
            msg = f"cannot load {p}: {exc}"
            raise EngineUnavailableError(msg) from exc
        if lib.af_abi_version() != _abi.AF_ABI_VERSION:
            msg = "libasyncflow_hip.so was built against another ABI version; rebuild it"
            raise EngineUnavailableError(msg)
        _lib = _abi.declare(lib)
    return _lib


def _check(lib: C.CDLL, rc: int, what: str) -> None:
    if rc == _abi.AF_OK:
        return
    text = (lib.af_last_error() or b"").decode(errors="replace")
    if rc == _abi.AF_ERR_NO_DEVICE:
        raise EngineUnavailableError(f"{what}: {text}")
    raise EngineError(f"{what} failed ({rc}): {text}")


def flow_mode(flow: bool | str) -> int:
    """`af_engine_options_t.flow_mode` of the `flow=` keyword: True = the engine's choice (0), False = never (1),
    "always" = whenever the plan is in the stage-parallel kernel's range (2)."""
    if flow == "always":
        return 2
    if isinstance(flow, str):
        msg = f"flow must be True, False or 'always', not {flow!r}"
        raise ValueError(msg)
    return 0 if flow else 1


PLAN_ONLY = _abi.DEVICE_PLAN_ONLY   # Engine(plan, PLAN_ONLY): a planning-only engine (include/asyncflow_hip.h, AF_DEVICE_PLAN_ONLY)


class Engine:
    """One ``af_engine_t``: a lowered plan resident on one GPU."""

    def __init__(self, plan: DevicePlan, device: int = 0, *, request_capacity: int = 0,
                 fifo_capacity: int = 0, force_global_state: bool = False, lanes_per_wave: int = 0,
                 draw_memory_mb: int = 0, expect_shared_instants: bool = False, flow: bool | str = True,
                 flow_list_entries: int = 0, flow_ring_rows: int = 0) -> None:
        self._lib = load_library()
        self.plan = plan
        self.device = device
        self._cplan = plan.as_ctypes()
        opts = _abi.AfEngineOptions(request_capacity, fifo_capacity, int(force_global_state), int(lanes_per_wave),
                                    int(draw_memory_mb), int(expect_shared_instants), flow_mode(flow),
                                    int(flow_list_entries), int(flow_ring_rows))
        handle = C.c_void_p()
        _check(self._lib, self._lib.af_engine_create(C.byref(self._cplan), device, C.byref(opts), C.byref(handle)),
               "af_engine_create")
        self._h = handle
        self._jit_spec: bytes | None = None
        self._jit_image: bytes | None = None

    def _sweep_structs(self, seeds, overrides, clock_ptr, clock_capacity, samples_ptr, tick_capacity, counts_ptr,
                       draw_capacity, online_hist_ptr, online_hist_bins, online_hist_max, online_rps_ptr,
                       online_rps_buckets):
        seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
        n = int(seeds.shape[0])
        cols = [np.ascontiguousarray(v, dtype=np.float64) for _, _, v in overrides]
        for c in cols:
            if c.shape != (n,):
                msg = f"override column has shape {c.shape}, expected ({n},)"
                raise ValueError(msg)
        ov = (_abi.AfOverride * max(len(cols), 1))()
        for k, (param, index, _) in enumerate(overrides):
            ov[k].param, ov[k].index = int(param), int(index)
            ov[k].values = cols[k].ctypes.data_as(C.POINTER(C.c_double))
        sweep = _abi.AfSweep(n, seeds.ctypes.data_as(C.POINTER(C.c_uint64)), len(cols), ov, int(draw_capacity))
        out = _abi.AfOutputs(int(clock_capacity), C.c_void_p(clock_ptr or None), int(tick_capacity),
                             C.c_void_p(samples_ptr or None), C.c_void_p(counts_ptr),
                             int(online_hist_bins), float(online_hist_max), C.c_void_p(online_hist_ptr or None),
                             int(online_rps_buckets), C.c_void_p(online_rps_ptr or None))
        return sweep, out, (seeds, cols, ov)     # the last item keeps the host arrays alive

    def run(self, seeds: np.ndarray, overrides: Sequence[tuple[int, int, np.ndarray]], *,
            clock_ptr: int, clock_capacity: int, samples_ptr: int, tick_capacity: int, counts_ptr: int,
            draw_capacity: int = 0, specialise: bool = False, online_hist_ptr: int = 0, online_hist_bins: int = 0,
            online_hist_max: float = 0.0, online_rps_ptr: int = 0, online_rps_buckets: int = 0,
            specialise_build: bool = True, summary: dict | None = None) -> _abi.AfStats:
        """Launch the sweep; output pointers are DEVICE addresses owned by the caller.

        ``specialise``: build (or fetch from the cache) kernels with this plan's shape as compile-time
        constants and use them for this sweep (asyncflow_amd/jit.py; worth it for long sweeps);
        ``specialise_build=False``: only from the cache, never a hipcc run.
        ``summary``: the keyword arguments of :meth:`summarize` that name its outputs (``stats_ptr``, ``rps_ptr``,
        ``rps_buckets``, ``hist_ptr``, ``hist_bins``, ``hist_max``, ``series_mean_ptr``, ``series_max_ptr``): run and
        analyzer in ONE call (``af_engine_run_summarized``) -- the same results as ``run`` then ``summarize``, with the
        analyzer of the stage-parallel kernel's full residency rounds hidden beside its last, partial one.
        """
        sweep, out, _keep = self._sweep_structs(seeds, overrides, clock_ptr, clock_capacity, samples_ptr, tick_capacity,
                                                counts_ptr, draw_capacity, online_hist_ptr, online_hist_bins,
                                                online_hist_max, online_rps_ptr, online_rps_buckets)
        if specialise:
            self._specialise(sweep, out, build=specialise_build)
        if summary is not None:
            summ = _abi.AfSummary(int(seeds.shape[0]) if hasattr(seeds, "shape") else len(seeds), int(summary.get("rps_buckets", 0)),
                                  int(summary.get("hist_bins", 0)), float(summary.get("hist_max", 0.0)),
                                  C.c_void_p(summary.get("stats_ptr") or None), C.c_void_p(summary.get("rps_ptr") or None),
                                  C.c_void_p(summary.get("hist_ptr") or None), C.c_void_p(summary.get("series_mean_ptr") or None),
                                  C.c_void_p(summary.get("series_max_ptr") or None))
            _check(self._lib, self._lib.af_engine_run_summarized(self._h, C.byref(sweep), C.byref(out), C.byref(summ)),
                   "af_engine_run_summarized")
            return self.stats()
        _check(self._lib, self._lib.af_engine_run(self._h, C.byref(sweep), C.byref(out)), "af_engine_run")
        return self.stats()

    def prepare(self, seeds: np.ndarray, overrides: Sequence[tuple[int, int, np.ndarray]], **kw) -> None:
        """Build / load the plan-specialised kernels for a sweep of this shape without running it
        (same keyword arguments as :meth:`run`)."""
        kw.pop("specialise", None)
        defaults = {"draw_capacity": 0, "online_hist_ptr": 0, "online_hist_bins": 0, "online_hist_max": 0.0,
                    "online_rps_ptr": 0, "online_rps_buckets": 0}
        defaults.update(kw)
        sweep, out, _keep = self._sweep_structs(seeds, overrides, defaults["clock_ptr"], defaults["clock_capacity"],
                                                defaults["samples_ptr"], defaults["tick_capacity"], defaults["counts_ptr"],
                                                defaults["draw_capacity"], defaults["online_hist_ptr"],
                                                defaults["online_hist_bins"], defaults["online_hist_max"],
                                                defaults["online_rps_ptr"], defaults["online_rps_buckets"])
        self._specialise(sweep, out)

    def jit_spec(self, seeds: np.ndarray, overrides: Sequence[tuple[int, int, np.ndarray]], **kw) -> str:
        """The ``-D`` flags of the plan-specialised kernels a sweep of this shape launches (same keyword arguments as
        :meth:`run`; of the pointers only their PRESENCE enters).  Works on a planning-only engine
        (``device=PLAN_ONLY``: no GPU needed) for sweeps the stage-parallel kernel runs."""
        kw.pop("specialise", None)
        defaults = {"draw_capacity": 0, "online_hist_ptr": 0, "online_hist_bins": 0, "online_hist_max": 0.0,
                    "online_rps_ptr": 0, "online_rps_buckets": 0}
        defaults.update(kw)
        sweep, out, _keep = self._sweep_structs(seeds, overrides, defaults["clock_ptr"], defaults["clock_capacity"],
                                                defaults["samples_ptr"], defaults["tick_capacity"], defaults["counts_ptr"],
                                                defaults["draw_capacity"], defaults["online_hist_ptr"],
                                                defaults["online_hist_bins"], defaults["online_hist_max"],
                                                defaults["online_rps_ptr"], defaults["online_rps_buckets"])
        return self._spec_of(sweep, out).decode()

    def _spec_of(self, sweep: _abi.AfSweep, out: _abi.AfOutputs) -> bytes:
        buf = C.create_string_buffer(2048)
        _check(self._lib, self._lib.af_engine_jit_spec(self._h, C.byref(sweep), C.byref(out), buf, len(buf)),
               "af_engine_jit_spec")
        return buf.value

    def _specialise(self, sweep: _abi.AfSweep, out: _abi.AfOutputs, build: bool = True) -> None:
        import warnings

        from . import jit

        spec = self._spec_of(sweep, out)
        if spec == self._jit_spec:
            return
        try:
            image = jit.code_object(spec.decode(), build=build)
        except jit.JitUnavailableError as exc:      # not an error: the generic kernels do the same job
            if not build:
                # too short a sweep to wait for the compiler: it runs on the generic kernels, and the build goes on beside it
                # for the next sweep of this shape (asyncflow_amd/jit.py::build_in_background)
                jit.build_in_background(spec.decode())
                return
            warnings.warn(f"plan-specialised kernels unavailable, using the generic ones: {exc}", RuntimeWarning,
                          stacklevel=3)
            return
        self._jit_image = image                     # keep the bytes alive while the module is loaded
        _check(self._lib, self._lib.af_engine_set_kernels(self._h, spec, image, len(image)), "af_engine_set_kernels")
        self._jit_spec = spec

    def summarize(self, n: int, *, clock_ptr: int, clock_capacity: int, samples_ptr: int, tick_capacity: int,
                  counts_ptr: int, stats_ptr: int = 0, rps_ptr: int = 0, rps_buckets: int = 0, hist_ptr: int = 0,
                  hist_bins: int = 0, hist_max: float = 0.0, series_mean_ptr: int = 0,
                  series_max_ptr: int = 0) -> _abi.AfStats:
        """Batched analyzer on the device over the outputs of a finished run (DEVICE pointers)."""
        out = _abi.AfOutputs(int(clock_capacity), C.c_void_p(clock_ptr or None), int(tick_capacity),
                             C.c_void_p(samples_ptr or None), C.c_void_p(counts_ptr))
        summ = _abi.AfSummary(int(n), int(rps_buckets), int(hist_bins), float(hist_max),
                              C.c_void_p(stats_ptr or None), C.c_void_p(rps_ptr or None), C.c_void_p(hist_ptr or None),
                              C.c_void_p(series_mean_ptr or None), C.c_void_p(series_max_ptr or None))
        _check(self._lib, self._lib.af_engine_summarize(self._h, C.byref(out), C.byref(summ)), "af_engine_summarize")
        return self.stats()

    def gather(self, comm: "C.c_void_p | int", world_size: int, n_local: int, local: dict, gathered: dict, *,
               rps_buckets: int = 0, hist_bins: int = 0) -> _abi.AfStats:
        """``af_engine_gather``: ONE grouped RCCL all-gather of the per-scenario summaries.

        ``local`` / ``gathered`` map ``stats | rps | hist | series_mean | series_max`` to DEVICE pointers
        (0 / missing = not gathered); every rank passes the same ``n_local`` (pad the last shard)."""
        def summ(n: int, ptrs: dict) -> _abi.AfSummary:
            return _abi.AfSummary(int(n), int(rps_buckets), int(hist_bins), 0.0,
                                  C.c_void_p(ptrs.get("stats") or None), C.c_void_p(ptrs.get("rps") or None),
                                  C.c_void_p(ptrs.get("hist") or None), C.c_void_p(ptrs.get("series_mean") or None),
                                  C.c_void_p(ptrs.get("series_max") or None))

        a, b = summ(n_local, local), summ(n_local * world_size, gathered)
        _check(self._lib, self._lib.af_engine_gather(self._h, comm, int(world_size), C.byref(a), C.byref(b)), "af_engine_gather")
        return self.stats()

    def flow_reason(self) -> str:
        """'' when the stage-parallel kernel can run this plan, else why it always runs on the next-event kernels."""
        return (self._lib.af_engine_flow_reason(self._h) or b"").decode()

    def stats(self) -> _abi.AfStats:
        st = _abi.AfStats()
        _check(self._lib, self._lib.af_engine_stats(self._h, C.byref(st)), "af_engine_stats")
        return st

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.af_engine_destroy(self._h)
            self._h = None

    def __del__(self) -> None:  # pragma: no cover - best effort
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


def probe_math(kind: int, x: np.ndarray, y: np.ndarray | None = None, seed: int = 0, device: int = 0) -> np.ndarray:
    """Evaluate the engine's own RNG/math on the GPU (spec pinning, tests only)."""
    lib = load_library()
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    pd = C.POINTER(C.c_double)
    yp = None
    if y is not None:
        y = np.ascontiguousarray(y, dtype=np.float64)
        yp = y.ctypes.data_as(pd)
    _check(lib, lib.af_probe_math(device, kind, C.c_uint64(seed), x.ctypes.data_as(pd), yp, out.ctypes.data_as(pd), x.size),
           "af_probe_math")
    return out
