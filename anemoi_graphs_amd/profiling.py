"""Per-kernel device timing from HIP events the library records on its launch stream."""
from __future__ import annotations

import ctypes

from . import _lib


def enable(on: bool = True) -> None:
    _lib.check(_lib.load().ag_profile_enable(int(on)), "ag_profile_enable")


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
