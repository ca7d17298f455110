"""Loader of the product library ``libepsm_hip.so`` (C ABI: include/epsm.h).

There is no CPU fallback: if the library is missing or a call fails, an
exception is raised.  ``import torch`` happens first on purpose -- torch-ROCm
ships its own ``libamdhip64.so.7`` and the dynamic loader then binds our
library to that same HIP runtime, so torch streams / device pointers are valid
inside our kernels.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import torch  # noqa: F401  (must be loaded before libepsm_hip.so, see above)

_HERE = os.path.dirname(os.path.abspath(__file__))
ABI_VERSION = 7          # EPSM_ABI_VERSION of include/epsm.h
LIB_PATH = os.path.join(_HERE, os.environ.get("EPSM_LIB_NAME", "libepsm_hip.so"))   # EPSM_LIB_NAME: A/B builds
_lib = None


class EpsmError(RuntimeError):
    pass


class build_lock:
    """One build of a directory at a time across processes (the ranks of a multi-process run all check their libraries on
    start-up; two `make`s in one directory write the same files, and a third process may dlopen a half-written one)."""

    def __init__(self, directory: str):
        self.path = os.path.join(directory, ".build.lock")

    def __enter__(self):
        import fcntl
        self.f = open(self.path, "w")
        fcntl.flock(self.f, fcntl.LOCK_EX)
        return self

    def __exit__(self, *exc):
        import fcntl
        fcntl.flock(self.f, fcntl.LOCK_UN)
        self.f.close()
        return False


def build(force: bool = False) -> str:
    """Compiles the HIP library for gfx950 (hipcc cross-compiles without a GPU)."""
    src_dir = os.path.join(_HERE, "csrc")
    with build_lock(src_dir):
        if force:
            subprocess.run(["make", "-C", src_dir, "-s", "clean"], check=True)
        subprocess.run(["make", "-C", src_dir, "-s"], check=True)
    return LIB_PATH


def _declare(lib):
    lib.epsm_abi_version.restype = C.c_int
    lib.epsm_abi_version.argtypes = []
    lib.epsm_last_error.restype = C.c_char_p
    lib.epsm_last_error.argtypes = []
    lib.epsm_num_param_grads.restype = C.c_int
    lib.epsm_num_param_grads.argtypes = [C.c_int, C.c_int]
    lib.epsm_manifold_grad.restype = C.c_int
    lib.epsm_manifold_grad.argtypes = [
        C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p,
        C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_float,
        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.epsm_first_vertex_tangent.restype = C.c_int
    lib.epsm_first_vertex_tangent.argtypes = [
        C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
        C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
        C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.epsm_scatter.restype = C.c_int
    lib.epsm_scatter.argtypes = [
        C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
        C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
    lib.epsm_manifold_grad_scatter.restype = C.c_int
    lib.epsm_manifold_grad_scatter.argtypes = [
        C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
        C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_float,
        C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
    lib.epsm_backward_pass.restype = C.c_int
    lib.epsm_backward_pass.argtypes = [
        C.c_int, C.c_int64, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
        C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float,
        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
    lib.epsm_backward_pass_packed.restype = C.c_int
    lib.epsm_backward_pass_packed.argtypes = [
        C.c_int, C.c_int64, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
        C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
    lib.epsm_sinkhorn_splits.restype = C.c_int
    lib.epsm_sinkhorn_splits.argtypes = [C.c_int64, C.c_int64]
    lib.epsm_sinkhorn_scratch_bytes.restype = C.c_size_t
    lib.epsm_sinkhorn_scratch_bytes.argtypes = [C.c_int64, C.c_int64, C.c_int]
    lib.epsm_sinkhorn_softmin.restype = C.c_int
    lib.epsm_sinkhorn_softmin.argtypes = [C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.epsm_sinkhorn_update.restype = C.c_int
    lib.epsm_sinkhorn_update.argtypes = [C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.epsm_release_workspace.restype = C.c_int
    lib.epsm_release_workspace.argtypes = []
    lib.epsm_set_option.restype = C.c_int
    lib.epsm_set_option.argtypes = [C.c_int, C.c_int64]
    lib.epsm_get_option.restype = C.c_int64
    lib.epsm_get_option.argtypes = [C.c_int]
    declare_tracer(lib)
    return lib


def declare_tracer(lib):
    """Prototypes of include/epsm_trace.h (also applied to the host build of the tracer in tests/host_harness)."""
    trace_args = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int,
                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                  C.c_void_p, C.c_uint32]
    lib.epsm_trace_paths.restype = C.c_int
    lib.epsm_trace_paths.argtypes = trace_args + [C.c_void_p]
    lib.epsm_trace_paths_wavefront.restype = C.c_int
    lib.epsm_trace_paths_wavefront.argtypes = trace_args + [C.c_void_p, C.c_size_t, C.c_void_p]
    lib.epsm_trace_paths_color.restype = C.c_int
    lib.epsm_trace_paths_color.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.epsm_trace_workspace_bytes.restype = C.c_size_t
    lib.epsm_trace_workspace_bytes.argtypes = [C.c_int64]
    lib.epsm_film_splat.restype = C.c_int
    lib.epsm_film_splat.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.epsm_film_adjoint_reparam.restype = C.c_int
    lib.epsm_film_adjoint_reparam.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                              C.c_void_p, C.c_void_p, C.c_void_p]
    lib.epsm_film_develop.restype = C.c_int
    lib.epsm_film_develop.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    return lib


def lib():
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise EpsmError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or make -C epsm_mitsuba3_amd/csrc). There is no CPU fallback.")
        _lib = _declare(C.CDLL(LIB_PATH))
        if _lib.epsm_abi_version() != ABI_VERSION:
            raise EpsmError("libepsm_hip.so ABI version mismatch")
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().epsm_last_error().decode("utf-8", "replace")
        raise EpsmError(f"{what} failed with code {rc}: {msg}")


OPT_SMALL_WAVEFRONT_PATHS, OPT_REPLICAS, OPT_ONE_LAUNCH = 0, 1, 2       # include/epsm.h EPSM_OPT_*


class options:
    """``with options(small_wavefront_paths=0, replicas=False): ...`` -- launch options of the fused entry points
    (include/epsm.h, epsm_set_option) for the duration of a block; the previous values come back on exit."""

    def __init__(self, small_wavefront_paths=None, replicas=None, one_launch=None):
        self.want = {OPT_SMALL_WAVEFRONT_PATHS: small_wavefront_paths, OPT_REPLICAS: None if replicas is None else int(bool(replicas)),
                     OPT_ONE_LAUNCH: None if one_launch is None else int(bool(one_launch))}

    def __enter__(self):
        self.saved = {k: lib().epsm_get_option(k) for k, v in self.want.items() if v is not None}
        for k, v in self.want.items():
            if v is not None:
                check(lib().epsm_set_option(k, int(v)), "epsm_set_option")
        return self

    def __exit__(self, *exc):
        for k, v in self.saved.items():
            lib().epsm_set_option(k, v)
        return False
