"""Per-kernel device timing from HIP events the library records on its launch stream."""
from __future__ import annotations

import ctypes

from . import _lib


def enable(on: bool = True, only: str | None = None) -> None:
    """Start (or stop) bracketing kernel launches with events; ``only`` restricts the events to
    launches of one kernel name, which keeps the host cost out of a timed region."""
    lib = _lib.load()
    _lib.check(lib.ag_profile_filter(only.encode() if only else None), "ag_profile_filter")
    _lib.check(lib.ag_profile_enable(int(on)), "ag_profile_enable")


def collect() -> dict[str, tuple[int, float]]:
    """{kernel name: (launches, total milliseconds)} since ``enable(True)``; synchronises."""
    lib = _lib.load()
    buf = ctypes.create_string_buffer(1 << 16)
    lib.ag_profile_collect(buf, len(buf))
    out = {}
    for line in buf.value.decode().splitlines():
        name, launches, ms = line.rsplit(" ", 2)
        out[name] = (int(launches), float(ms))
    return out
