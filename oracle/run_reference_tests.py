"""Run the reference's OWN test-suite against the SimPy stand-in (conformance).

ORACLE / TEST INFRASTRUCTURE ONLY.  Build-container only (needs /root/reference).

    python oracle/run_reference_tests.py            # all 183 tests, incl. system
    python oracle/run_reference_tests.py -k server  # pytest args are forwarded

The reference tests are what pins the stand-in kernel:
tests/unit/runtime/events/test_injection_*.py (peek/step/zero-time ordering),
tests/unit/runtime/actors/test_server.py (Container + run(until) timing),
tests/unit/runtime/test_simulation_runner.py:183-195 (Store FIFO), ...
"""

from __future__ import annotations

import os
import sys
import tempfile
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from oracle import ref_env  # noqa: E402


def main(argv: list[str]) -> int:
    ref_env.install()
    import pytest

    os.environ.setdefault("ASYNCFLOW_RUN_SYSTEM_TESTS", "1")
    os.environ.setdefault("MPLBACKEND", "Agg")
    root = ref_env.REFERENCE_ROOT
    with tempfile.TemporaryDirectory() as cwd:
        os.chdir(cwd)
        args = [
            "-p", "no:cacheprovider",
            "-o", "addopts=",
            "-q",
            f"--rootdir={root}",
            "-c", str(root / "pytest.ini"),
            str(root / "tests"),
            *argv,
        ]
        return int(pytest.main(args))


if __name__ == "__main__":
    raise SystemExit(main(sys.argv[1:]))
