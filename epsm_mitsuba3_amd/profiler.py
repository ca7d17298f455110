"""Stage markers for profilers: the counterpart of the reference's ``ScopedPhase`` (include/mitsuba/core/profiler.h:88-96,
which forwards its ``ProfilerPhase`` names to NVTX / ITT when built with MI_PROFILER_NVTX / _ITTNOTIFY,
CMakeLists.txt:42-43, 407-412).  Here the ranges go to ROCTX (``libroctx64`` / ``librocprofiler-sdk-roctx``), which
``rocprofv3 --marker-trace`` records; without the library -- or with ``EPSM_ROCTX=0`` -- every call is a no-op.

    with profiler.phase("trace"):
        ...
"""
from __future__ import annotations

import ctypes as C
import os

_roctx = None
_tried = False


def _lib():
    global _roctx, _tried
    if not _tried:
        _tried = True
        if os.environ.get("EPSM_ROCTX", "1") != "0":
            for name in ("librocprofiler-sdk-roctx.so", "libroctx64.so"):
                try:
                    lib = C.CDLL(name)
                    lib.roctxRangePushA.restype = C.c_int
                    lib.roctxRangePushA.argtypes = [C.c_char_p]
                    lib.roctxRangePop.restype = C.c_int
                    lib.roctxRangePop.argtypes = []
                    _roctx = lib
                    break
                except (OSError, AttributeError):
                    continue
    return _roctx


def available() -> bool:
    return _lib() is not None


class phase:
    """A named range around one stage of ``render`` / ``render_backward`` (trace, backward pass, all-reduce, ...)."""
    __slots__ = ("name", "on")

    def __init__(self, name: str):
        self.name = name
        self.on = False

    def __enter__(self):
        lib = _lib()
        if lib is not None:
            lib.roctxRangePushA(self.name.encode())
            self.on = True
        return self

    def __exit__(self, *exc):
        if self.on:
            _roctx.roctxRangePop()
        return False
