"""Compile the HIP engine for gfx950 into the in-tree shared library.

    python -m asyncflow_amd.build            # hipcc cross-compiles without a GPU

The library is kept IN-TREE (asyncflow_amd/csrc/libasyncflow_hip.so): it is
git-ignored but travels with the repository snapshot to the GPU box.
"""

from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
from pathlib import Path

CSRC = Path(__file__).resolve().parent / "csrc"
LIB_PATH = CSRC / "libasyncflow_hip.so"
SOURCES = ("engine.hip", "af_flow.hpp", "af_flow_host.hpp", "af_core.hpp", "af_math.hpp", "af_plan_pack.hpp", "af_summary.hpp", "af_pregen.hpp")
ARCH = "gfx950"

# -ffp-contract=off / -fno-fast-math: every f64 expression is evaluated exactly as
# written (no FMA contraction), which is what makes device results reproducible
# bit for bit by the CPU oracle.
HIPCC_FLAGS = (
    f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function",
    # Do NOT record a dependency on a particular libamdhip64: the HIP runtime is
    # resolved at load time from the one already in the process (PyTorch bundles
    # its own copy; two HIP/HSA runtimes in one process do not coexist reliably).
    "-no-hip-rt",
)


def hipcc_path() -> str:
    if os.environ.get("ASYNCFLOW_NO_HIPCC"):      # (tests / measurements: behave like a box without a compiler)
        msg = "hipcc disabled by ASYNCFLOW_NO_HIPCC"
        raise RuntimeError(msg)
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    msg = "hipcc not found (set HIPCC or install ROCm)"
    raise RuntimeError(msg)


STAMP_PATH = CSRC / "libasyncflow_hip.so.stamp"   # sha1 of the sources the library was built from (git-ignored, travels to the GPU box)
INCLUDE = CSRC.parent.parent / "include" / "asyncflow_hip.h"


def sources_sha1() -> str:
    """Content hash of everything the library is compiled from (mtimes do not survive a snapshot copy)."""
    h = hashlib.sha1()  # noqa: S324 - a change detector, not a security boundary
    for path in [*(CSRC / s for s in SOURCES), INCLUDE]:
        h.update(path.name.encode())
        h.update(path.read_bytes())
    h.update(" ".join(HIPCC_FLAGS).encode())
    return h.hexdigest()


def needs_build() -> bool:
    """True when the library is missing or was built from other sources than the ones in the tree."""
    if not LIB_PATH.exists() or not STAMP_PATH.exists():
        return True
    return STAMP_PATH.read_text().strip() != sources_sha1()


def have_hipcc() -> bool:
    try:
        hipcc_path()
    except RuntimeError:
        return False
    return True


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile the library unless it is up to date.  Safe to call from several processes at once (pytest-xdist,
    one rank per GPU): an exclusive file lock serialises the builds and the late comers find the stamp current."""
    if not force and not needs_build():
        return LIB_PATH
    import fcntl

    try:
        lock = open(CSRC / ".build.lock", "w")
    except OSError:
        # a read-only install: the lock lives in the temp directory (the build below then fails on its own, with hipcc's message)
        import tempfile

        lock = open(Path(tempfile.gettempdir()) / f"asyncflow_amd_build_{os.getuid()}.lock", "w")
    with lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not needs_build():
            return LIB_PATH
        want = sources_sha1()
        tmp = LIB_PATH.with_suffix(f".so.tmp{os.getpid()}")
        cmd = [hipcc_path(), *HIPCC_FLAGS, "-o", str(tmp), str(CSRC / "engine.hip")]
        res = subprocess.run(cmd, capture_output=True, text=True, check=False)
        if verbose or res.returncode != 0:
            print(" ".join(cmd))
            print(res.stdout, res.stderr)
        if res.returncode != 0:
            tmp.unlink(missing_ok=True)
            msg = f"hipcc failed ({res.returncode}):\n{res.stderr[-4000:]}"
            raise RuntimeError(msg)
        os.replace(tmp, LIB_PATH)      # (a process that has the old library mapped keeps its inode)
        STAMP_PATH.write_text(want + "\n")
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
