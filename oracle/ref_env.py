"""Make the UNMODIFIED reference importable in the build container.

ORACLE / TEST INFRASTRUCTURE ONLY -- the product path never imports this.

The reference targets Python >= 3.11 and SimPy 4.1.1; this image has Python
3.10 and no SimPy.  Three substitutions:

1. ``enum.StrEnum`` (used at /root/reference/src/asyncflow/config/constants.py:16)
   -> ``class StrEnum(str, Enum)`` whose ``__str__``/``__format__`` are ``str``'s.
2. ``typing.Self`` (used at .../builder/asyncflow_builder.py:5) ->
   ``typing_extensions.Self``.
3. ``simpy`` -> ``oracle/simpy_standin/simpy`` (restatement of SimPy 4.1.1).

``/root/reference`` exists only in the build container: callers must check
``reference_available()`` and skip otherwise (nothing on the GPU box may need it).
"""

from __future__ import annotations

import enum
import os
import sys
import typing
from pathlib import Path

REFERENCE_ROOT = Path(os.environ.get("ASYNCFLOW_REFERENCE_ROOT", "/root/reference"))
_STANDIN = Path(__file__).resolve().parent / "simpy_standin"


def reference_available() -> bool:
    return (REFERENCE_ROOT / "src" / "asyncflow" / "__init__.py").is_file()


def install(*, need_reference: bool = True) -> None:
    """Install shims + sys.path entries (idempotent)."""
    if not hasattr(enum, "StrEnum"):

        class StrEnum(str, enum.Enum):  # noqa: D401 - 3.11 backport
            """Backport of enum.StrEnum (3.11)."""

            def __new__(cls, value: str, *args: object) -> "StrEnum":
                if not isinstance(value, str):
                    msg = f"{value!r} is not a string"
                    raise TypeError(msg)
                member = str.__new__(cls, value)
                member._value_ = value
                return member

            __str__ = str.__str__
            __format__ = str.__format__  # type: ignore[assignment]

            @staticmethod
            def _generate_next_value_(name, start, count, last_values):  # type: ignore[override]
                return name.lower()

        enum.StrEnum = StrEnum  # type: ignore[attr-defined]

    if not hasattr(typing, "Self"):
        import typing_extensions

        typing.Self = typing_extensions.Self  # type: ignore[attr-defined]

    try:
        import simpy  # noqa: F401  (a real SimPy wins if one is ever installed)
    except ImportError:
        if str(_STANDIN) not in sys.path:
            sys.path.insert(0, str(_STANDIN))

    if need_reference:
        if not reference_available():
            msg = f"reference sources not found under {REFERENCE_ROOT}"
            raise RuntimeError(msg)
        src = str(REFERENCE_ROOT / "src")
        if src not in sys.path:
            sys.path.insert(0, src)


def simpy_flavour() -> str:
    """'real' if a genuine SimPy is importable, else 'standin'."""
    install(need_reference=False)
    import simpy

    return "standin" if "standin" in getattr(simpy, "__version__", "") else "real"
