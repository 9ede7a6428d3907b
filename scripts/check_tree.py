#!/usr/bin/env python3
"""Gate against a tree that cannot build (round 3 ended on a header cut off at 96 KiB).

    python scripts/check_tree.py [--staged] [--no-compile]

* every native source (asyncflow_amd/csrc/*, include/*.h, tests/hostcheck/*.cpp|hpp, oracle/*.c|h) ends in a newline,
  has balanced () [] {} outside comments and literals, and is not an exact multiple of 32 KiB (a cut-off write);
* `hipcc -fsyntax-only --offload-arch=gfx950` of engine.hip (device and host pass), `g++ -fsyntax-only` of the
  host-check build and `gcc -fsyntax-only` of the C oracle parse.
Installed as .git/hooks/pre-commit by scripts/install_hooks.sh; tests/test_00_tree.py runs the same checks first.
With --staged the files are read from the git index (what the commit will hold), not from the working tree.
"""
from __future__ import annotations

import re
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
GLOBS = ("asyncflow_amd/csrc/*.hip", "asyncflow_amd/csrc/*.hpp", "include/*.h", "tests/hostcheck/*.cpp", "tests/hostcheck/*.hpp",
         "oracle/*.c", "oracle/*.h")
_STRIP = re.compile(r'//[^\n]*|/\*.*?\*/|"(?:\\.|[^"\\\n])*"|\'(?:\\.|[^\'\\\n])*\'', re.S)


def native_sources(root: Path = ROOT) -> list[Path]:
    return sorted(p for g in GLOBS for p in root.glob(g))


def first_branches(code: str) -> str:
    """Of every #if / #elif / #else group keep the FIRST branch only (the branches of one group close the same
    braces); directive lines themselves become empty lines, so line numbers stay."""
    out, skip = [], []          # skip[k]: the k-th open conditional is past its first branch
    for line in code.split("\n"):
        s = line.lstrip()
        if s.startswith("#"):
            word = s[1:].lstrip()
            if word.startswith("if"):
                skip.append(False)
            elif word.startswith(("elif", "else")) and skip:
                skip[-1] = True
            elif word.startswith("endif") and skip:
                skip.pop()
            out.append("")
        else:
            out.append("" if any(skip) else line)
    return "\n".join(out)


def text_problems(name: str, data: bytes) -> list[str]:
    out = []
    if not data:
        return [f"{name}: empty"]
    if not data.endswith(b"\n"):
        out.append(f"{name}: no newline at the end of the file (cut off?)")
    if len(data) % 32768 == 0:
        out.append(f"{name}: {len(data)} bytes is an exact multiple of 32 KiB (a cut-off write?)")
    code = first_branches(_STRIP.sub(" ", data.decode("utf-8", errors="replace")))
    pairs = {")": "(", "]": "[", "}": "{"}
    stack: list[tuple[str, int]] = []
    line = 1
    for ch in code:
        if ch == "\n":
            line += 1
        elif ch in "([{":
            stack.append((ch, line))
        elif ch in pairs:
            if not stack or stack[-1][0] != pairs[ch]:
                out.append(f"{name}:{line}: unmatched '{ch}'")
                return out
            stack.pop()
    if stack:
        out.append(f"{name}:{stack[-1][1]}: '{stack[-1][0]}' is never closed ({len(stack)} open at the end of the file)")
    return out


def compile_problems(root: Path = ROOT) -> list[str]:
    out = []
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if Path(hipcc).exists():
        r = subprocess.run([hipcc, "-fsyntax-only", "--offload-arch=gfx950", "-std=c++17", "-Wno-unused-command-line-argument",
                            str(root / "asyncflow_amd/csrc/engine.hip")], capture_output=True, text=True, check=False)
        if r.returncode != 0:
            out.append("hipcc -fsyntax-only engine.hip failed:\n" + r.stderr[-3000:])
    else:
        out.append("hipcc not found: engine.hip not parsed")
    r = subprocess.run(["g++", "-fsyntax-only", "-std=c++17", str(root / "tests/hostcheck/hostcheck.cpp")],
                       capture_output=True, text=True, check=False)
    if r.returncode != 0:
        out.append("g++ -fsyntax-only hostcheck.cpp failed:\n" + r.stderr[-3000:])
    r = subprocess.run(["gcc", "-fsyntax-only", str(root / "oracle/des_oracle.c")], capture_output=True, text=True, check=False)
    if r.returncode != 0:
        out.append("gcc -fsyntax-only des_oracle.c failed:\n" + r.stderr[-3000:])
    return out


def staged_tree() -> Path:
    """The git index checked out into a scratch directory (what the commit will contain)."""
    tmp = Path(tempfile.mkdtemp(prefix="af_staged_"))
    subprocess.run(["git", "checkout-index", "-a", f"--prefix={tmp}/"], cwd=ROOT, check=True)
    return tmp


def main() -> int:
    root = ROOT
    scratch = None
    if "--staged" in sys.argv:
        root = scratch = staged_tree()
    try:
        srcs = native_sources(root)
        problems = [msg for p in srcs for msg in text_problems(str(p.relative_to(root)), p.read_bytes())]
        if "--no-compile" not in sys.argv and not problems:
            problems += compile_problems(root)
    finally:
        if scratch is not None:
            shutil.rmtree(scratch, ignore_errors=True)
    for msg in problems:
        print("check_tree:", msg, file=sys.stderr)
    if not problems:
        print(f"check_tree: {len(srcs)} native sources are whole and parse")
    return 1 if problems else 0


if __name__ == "__main__":
    raise SystemExit(main())
