"""pytest configuration: `gpu` marker + repo root on sys.path."""

from __future__ import annotations

import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN_DIR = ROOT / "tests" / "golden"


# the suites run hundreds of short sweeps of unlike plans: no compiler thread behind each of them (the two tests of the
# background build switch it on for themselves)
import os  # noqa: E402

os.environ.setdefault("ASYNCFLOW_JIT_BACKGROUND", "0")


def pytest_configure(config: pytest.Config) -> None:
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


def golden_names() -> list[str]:
    # pooled_*: statistical fixture (round 5's deviation_* fixture is an ordinary parity fixture since round 6)
    return sorted(p.stem for p in GOLDEN_DIR.glob("*.npz") if not p.stem.startswith("pooled_"))
