"""Runs FIRST (file name): the tree's native sources are whole and parse (scripts/check_tree.py).

Round 3 ended on a header cut off in the middle of a function; every later test then failed for a reason that had
nothing to do with what it tests.  With `-x` a broken tree now stops here, with the file and line in the message.
"""

from __future__ import annotations

import importlib.util
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
_spec = importlib.util.spec_from_file_location("check_tree", ROOT / "scripts" / "check_tree.py")
check_tree = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(check_tree)


def test_native_sources_are_whole():
    srcs = check_tree.native_sources()
    assert len(srcs) >= 12
    problems = [m for p in srcs for m in check_tree.text_problems(str(p.relative_to(ROOT)), p.read_bytes())]
    assert not problems, "\n".join(problems)


def test_native_sources_parse():
    problems = check_tree.compile_problems()
    assert not problems, "\n".join(problems)


def test_the_gate_sees_a_cut_off_file():
    data = (ROOT / "asyncflow_amd" / "csrc" / "af_flow.hpp").read_bytes()
    found = check_tree.text_problems("af_flow.hpp", data[: 96 * 1024])
    assert any("never closed" in m or "unmatched" in m for m in found)
    assert any("32 KiB" in m for m in found)
