"""The C-ABI library loads on a CPU-only box and exports what include/asyncflow_hip.h declares."""

from __future__ import annotations

import ctypes as C
import re
from pathlib import Path

import pytest

from asyncflow_amd import _abi
from asyncflow_amd import build as af_build
from asyncflow_amd.plan import lower
from oracle.scenarios import lb_two_servers

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def lib():
    af_build.build()
    from asyncflow_amd.engine import load_library

    return load_library()


def test_header_symbols_are_all_exported(lib):
    header = (ROOT / "include" / "asyncflow_hip.h").read_text()
    declared = set(re.findall(r"\b(af_[a-z_]+)\s*\(", header))
    declared -= {"af_engine_t"}
    assert declared == set(_abi.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name


def test_abi_version_and_pure_helpers(lib):
    assert lib.af_abi_version() == _abi.AF_ABI_VERSION
    assert lib.af_tick_count(0.05, 600.0) == 11999
    assert lib.af_tick_count(0.01, 50.0) == 5000
    plan = lower(lb_two_servers())
    assert lib.af_series_count(C.byref(plan.as_ctypes())) == plan.n_series == 12
    assert plan.tick_count == 11999


def test_create_rejects_foreign_abi_and_reports_errors(lib):
    plan = lower(lb_two_servers()).as_ctypes()
    plan.abi_version = 999
    h = C.c_void_p()
    rc = lib.af_engine_create(C.byref(plan), 0, None, C.byref(h))
    assert rc == _abi.AF_ERR_ABI and not h.value
    assert b"ABI" in lib.af_last_error()


def test_no_cpu_fallback_without_gpu(lib):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from asyncflow_amd.engine import EngineUnavailableError
    from asyncflow_amd.runner import SimulationRunner

    with pytest.raises(EngineUnavailableError):
        SimulationRunner(simulation_input=lb_two_servers(horizon=5)).run()
    h = C.c_void_p()
    rc = lib.af_engine_create(C.byref(lower(lb_two_servers()).as_ctypes()), 0, None, C.byref(h))
    assert rc == _abi.AF_ERR_NO_DEVICE


def test_product_never_imports_the_oracle():
    """asyncflow_amd must not reference oracle/ or tests/hostcheck (no hidden CPU path)."""
    for path in (ROOT / "asyncflow_amd").rglob("*"):
        if path.suffix in {".py", ".hip", ".hpp", ".h"}:
            text = path.read_text()
            assert "import oracle" not in text and "from oracle" not in text, path
            assert "libaf_oracle" not in text and "libaf_hostcheck" not in text, path
