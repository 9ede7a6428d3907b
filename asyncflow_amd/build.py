"""Compile the HIP engine for gfx950 into the in-tree shared library.

    python -m asyncflow_amd.build            # hipcc cross-compiles without a GPU

The library is kept IN-TREE (asyncflow_amd/csrc/libasyncflow_hip.so): it is
git-ignored but travels with the repository snapshot to the GPU box.
"""

from __future__ import annotations

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
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    msg = "hipcc not found (set HIPCC or install ROCm)"
    raise RuntimeError(msg)


def needs_build() -> bool:
    if not LIB_PATH.exists():
        return True
    include = CSRC.parent.parent / "include" / "asyncflow_hip.h"
    newest = max([(CSRC / s).stat().st_mtime for s in SOURCES] + [include.stat().st_mtime])
    return LIB_PATH.stat().st_mtime < newest


def build(force: bool = False, verbose: bool = False) -> Path:
    if not force and not needs_build():
        return LIB_PATH
    cmd = [hipcc_path(), *HIPCC_FLAGS, "-o", str(LIB_PATH), str(CSRC / "engine.hip")]
    res = subprocess.run(cmd, capture_output=True, text=True, check=False)
    if verbose or res.returncode != 0:
        print(" ".join(cmd))
        print(res.stdout, res.stderr)
    if res.returncode != 0:
        msg = f"hipcc failed ({res.returncode}):\n{res.stderr[-4000:]}"
        raise RuntimeError(msg)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
