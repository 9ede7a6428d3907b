"""Plan-specialised kernels: compile csrc/engine.hip once more with the plan's shape baked in.

The shared library ships generic kernels.  For long sweeps it pays to build the same source again
with the plan's constants as ``-D`` flags (``af_engine_jit_spec`` produces them for whatever the sweep
would launch): the stage-parallel kernel as ONE entry point ``af_flow_jit`` (instantiation, plan shape,
horizon / tick constants and the whole LDS layout as immediates: its ~100 wave-uniform launch
arguments no longer spill through VGPR lanes), or the next-event kernels as ``af_jit_lean /
af_jit_order3 / af_jit_order2`` (state offsets become instruction immediates, loops over edges /
servers / series unroll).  Results are bit-identical (tests/test_gpu_parity.py, tests/test_gpu_flow.py).

One ``hipcc --genco`` call (~4 s) per distinct spec; code objects are cached in
``$ASYNCFLOW_JIT_CACHE`` (default ``asyncflow_amd/csrc/_jit/``; ``~/.cache/asyncflow_amd/jit`` when the
package directory is read-only) keyed by the spec and the sources' contents.  Every failure on the
way -- no hipcc, an unwritable cache, a failed build -- is a :class:`JitUnavailableError`: the caller
falls back to the generic kernels, it never crashes a run.
"""

from __future__ import annotations

import hashlib
import os
import shlex
import subprocess
import tempfile
import threading
from pathlib import Path

from .build import ARCH, CSRC, hipcc_path

CACHE_DIR = Path(os.environ["ASYNCFLOW_JIT_CACHE"]) if os.environ.get("ASYNCFLOW_JIT_CACHE") else CSRC / "_jit"
_FALLBACK_CACHE_DIR = Path.home() / ".cache" / "asyncflow_amd" / "jit"
_SOURCES = ("engine.hip", "af_flow.hpp", "af_flow_host.hpp", "af_core.hpp", "af_math.hpp", "af_plan_pack.hpp", "af_summary.hpp", "af_pregen.hpp")
_FLAGS = (f"--offload-arch={ARCH}", "--genco", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
          "-Wno-unused-function",
          # the five stations of af_flow_jit are unrolled by pragma (af_flow.hpp, Flow::run): no size limit on that
          "-mllvm", "-pragma-unroll-threshold=1000000",
          # experiment / profiling hook, part of the cache key: e.g. -gline-tables-only for rocprofv3's PC sampling
          *os.environ.get("ASYNCFLOW_JIT_EXTRA_FLAGS", "").split())
_source_digest: str | None = None


class JitUnavailableError(RuntimeError):
    """hipcc is missing or the specialised build failed (the generic kernels are used instead)."""


def _sources_digest() -> str:
    global _source_digest  # noqa: PLW0603
    if _source_digest is None:
        h = hashlib.sha1()
        for name in _SOURCES:
            h.update((CSRC / name).read_bytes())
        h.update((CSRC.parent.parent / "include" / "asyncflow_hip.h").read_bytes())
        _source_digest = h.hexdigest()
    return _source_digest


def _writable_cache_dir() -> Path:
    for cand in (CACHE_DIR, _FALLBACK_CACHE_DIR):
        try:
            cand.mkdir(parents=True, exist_ok=True)
            with tempfile.NamedTemporaryFile(dir=cand, suffix=".probe"):
                pass
        except OSError:
            continue
        return cand
    msg = f"no writable cache directory for plan-specialised kernels ({CACHE_DIR}, {_FALLBACK_CACHE_DIR})"
    raise JitUnavailableError(msg)


def code_object(spec: str, build: bool = True) -> bytes:
    """The code object for ``spec`` (the ``-D`` flags from ``af_engine_jit_spec``), built on demand.

    ``build=False``: only a cached object is returned (a sweep too short to repay a hipcc run still
    takes the specialised kernels when an earlier, longer one left them in the cache)."""
    key = hashlib.sha1((_sources_digest() + "\n" + spec + "\n" + " ".join(_FLAGS)).encode()).hexdigest()
    for cand in (CACHE_DIR, _FALLBACK_CACHE_DIR):
        try:
            return (cand / f"{key}.hsaco").read_bytes()
        except OSError:
            continue
    if not build:
        raise JitUnavailableError("not in the cache (and this sweep is too short to repay a build)")
    try:
        hipcc = hipcc_path()
    except RuntimeError as exc:
        raise JitUnavailableError(str(exc)) from exc
    try:
        cache = _writable_cache_dir()
        path = cache / f"{key}.hsaco"
        with tempfile.NamedTemporaryFile(dir=cache, suffix=".tmp", delete=False) as tmp:
            tmp_path = Path(tmp.name)
        cmd = [hipcc, *_FLAGS, *shlex.split(spec), "-o", str(tmp_path), str(CSRC / "engine.hip")]
        res = subprocess.run(cmd, capture_output=True, text=True, check=False)
        if res.returncode != 0 or tmp_path.stat().st_size == 0:
            tmp_path.unlink(missing_ok=True)
            msg = f"hipcc --genco failed ({res.returncode}):\n{res.stderr[-2000:]}"
            raise JitUnavailableError(msg)
        os.replace(tmp_path, path)          # atomic: concurrent ranks may build the same object
        return path.read_bytes()
    except OSError as exc:                  # read-only install, full disk, hipcc not executable, ...
        raise JitUnavailableError(f"{type(exc).__name__}: {exc}") from exc


def _sweep_stale_tmp_files(older_than_s: float = 600.0) -> None:
    """Temporary files of builds that never finished (a process that exited under its compiler): removed once they are older
    than any build could be."""
    import time

    now = time.time()
    for cand in (CACHE_DIR, _FALLBACK_CACHE_DIR):
        try:
            for f in cand.glob("*.tmp"):
                if now - f.stat().st_mtime > older_than_s:
                    f.unlink(missing_ok=True)
        except OSError:
            continue


def _wait_for_background_builds() -> None:
    """atexit: give the background builds a few seconds to land in the cache (see `build_in_background`)."""
    import time

    try:
        budget = float(os.environ.get("ASYNCFLOW_JIT_EXIT_WAIT_S", "8"))
    except ValueError:
        budget = 8.0
    deadline = time.monotonic() + max(budget, 0.0)
    for t in list(_background.values()):
        left = deadline - time.monotonic()
        if left <= 0.0:
            break
        t.join(left)


_background: dict[str, threading.Thread] = {}
_background_lock = threading.Lock()
_background_one_at_a_time = threading.Semaphore(1)      # background builds take turns: one compiler beside the simulation, not twenty


def build_in_background(spec: str) -> threading.Thread | None:
    """Build the code object of ``spec`` into the cache on a daemon thread (once per spec and process) and return the thread.

    For sweeps too short to repay a synchronous hipcc run (~4 s): the sweep at hand runs on the library's generic kernels
    -- BASELINE config 2 takes 64 instead of 38 ms per 10 000 replicas on them -- while the compiler works beside it, and the NEXT
    sweep of the same shape (the usual Monte-Carlo loop: same plan, other seeds) finds the specialised kernel in the cache.
    ``ASYNCFLOW_JIT_BACKGROUND=0`` turns it off.  Failures are silent (the generic kernels do the same job).  A process that
    finishes its sweep before the compiler does (~4 s) waits for the build at exit, at most ``ASYNCFLOW_JIT_EXIT_WAIT_S``
    seconds (default 8; 0: do not wait) -- otherwise a short script would never leave its kernel in the cache and its next run
    would be as slow; a build cut off anyway leaves at most a ``*.tmp`` file, never a truncated code object (the
    cache entry is renamed into place), and stale ``*.tmp`` files are swept when the next background build starts."""
    if os.environ.get("ASYNCFLOW_JIT_BACKGROUND", "1") == "0":
        return None
    with _background_lock:
        t = _background.get(spec)
        if t is not None:
            return t
        if not _background:
            import atexit

            atexit.register(_wait_for_background_builds)
            _sweep_stale_tmp_files()

        def work() -> None:
            with _background_one_at_a_time:
                try:
                    code_object(spec, build=True)
                except JitUnavailableError:
                    pass

        t = threading.Thread(target=work, name="asyncflow-jit", daemon=True)
        _background[spec] = t
        t.start()
        return t
