"""Profiler ranges.  ``range_push/pop`` and the ``annotate`` context manager emit roctx
markers (visible to ``rocprofv3 --marker-trace``) through ``librocprofiler-sdk-roctx``
when it can be loaded, and are no-ops otherwise.  The reference has no tracing at all;
its only instrumentation is wall-clock meters (/root/reference/2.distributed.py:204-236).
Enable with ``PDA_TRACE=1`` (off by default so the hot loop pays nothing)."""
from __future__ import annotations

import contextlib
import ctypes
import os

_ENABLED = os.environ.get("PDA_TRACE", "0") == "1"
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = False
        for name in ("librocprofiler-sdk-roctx.so", "libroctx64.so"):
            try:
                l = ctypes.CDLL(name)
                l.roctxRangePushA.argtypes = [ctypes.c_char_p]
                l.roctxRangePushA.restype = ctypes.c_int
                l.roctxRangePop.restype = ctypes.c_int
                _LIB = l
                break
            except (OSError, AttributeError):
                continue
    return _LIB


def enable(on: bool = True) -> None:
    global _ENABLED
    _ENABLED = on


def range_push(name: str) -> None:
    if _ENABLED and _lib():
        _LIB.roctxRangePushA(name.encode())


def range_pop() -> None:
    if _ENABLED and _lib():
        _LIB.roctxRangePop()


@contextlib.contextmanager
def annotate(name: str):
    range_push(name)
    try:
        yield
    finally:
        range_pop()
