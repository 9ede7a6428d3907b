"""Plan-specialised kernels: compile csrc/engine.hip once more with the plan's shape baked in.

The shared library ships generic next-event kernels.  For long sweeps it pays to build the same
source again with the plan's constants as ``-D`` flags (``af_engine_jit_spec`` produces them): state
offsets become instruction immediates, loops over edges / servers / series unroll, branches on the
plan's shape disappear -- about 8 % less kernel time on the 10 000-replica LB-2 sweep.  Results are
bit-identical (tests/test_gpu_parity.py).

One ``hipcc --genco`` call (~4 s) per distinct spec; code objects are cached in
``asyncflow_amd/csrc/_jit/`` keyed by the spec and the sources' contents.
"""

from __future__ import annotations

import hashlib
import os
import shlex
import subprocess
import tempfile
from pathlib import Path

from .build import ARCH, CSRC, hipcc_path

CACHE_DIR = CSRC / "_jit"
_SOURCES = ("engine.hip", "af_core.hpp", "af_math.hpp", "af_plan_pack.hpp", "af_summary.hpp")
_FLAGS = (f"--offload-arch={ARCH}", "--genco", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
          "-Wno-unused-function")
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


def code_object(spec: str) -> bytes:
    """The code object for ``spec`` (the ``-D`` flags from ``af_engine_jit_spec``), built on demand."""
    key = hashlib.sha1((_sources_digest() + "\n" + spec + "\n" + " ".join(_FLAGS)).encode()).hexdigest()
    path = CACHE_DIR / f"{key}.hsaco"
    if path.exists():
        return path.read_bytes()
    try:
        hipcc = hipcc_path()
    except RuntimeError as exc:
        raise JitUnavailableError(str(exc)) from exc
    CACHE_DIR.mkdir(parents=True, exist_ok=True)
    with tempfile.NamedTemporaryFile(dir=CACHE_DIR, suffix=".tmp", delete=False) as tmp:
        tmp_path = Path(tmp.name)
    cmd = [hipcc, *_FLAGS, *shlex.split(spec), "-o", str(tmp_path), str(CSRC / "engine.hip")]
    res = subprocess.run(cmd, capture_output=True, text=True, check=False)
    if res.returncode != 0 or tmp_path.stat().st_size == 0:
        tmp_path.unlink(missing_ok=True)
        msg = f"hipcc --genco failed ({res.returncode}):\n{res.stderr[-2000:]}"
        raise JitUnavailableError(msg)
    os.replace(tmp_path, path)          # atomic: concurrent ranks may build the same object
    return path.read_bytes()
