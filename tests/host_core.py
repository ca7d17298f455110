"""Builds and binds tests/host_harness (the kernel core compiled for the CPU)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import torch

from epsm_mitsuba3_amd.records import PackedRecords, VARIANTS, num_param_grads

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_harness")
_SO = os.path.join(_DIR, "libpath_core_host.so")
_SRC = os.path.join(_DIR, "path_core_host.cpp")
_HDR = os.path.join(os.path.dirname(_DIR), "..", "epsm_mitsuba3_amd", "csrc", "epsm_path_core.h")
_lib = None


def lib():
    global _lib
    if _lib is None:
        stale = (not os.path.isfile(_SO)) or any(
            os.path.getmtime(p) > os.path.getmtime(_SO) for p in (_SRC, _HDR))
        if stale:
            subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas",
                            "-ffp-contract=off", "-o", _SO, _SRC], check=True)
        _lib = C.CDLL(_SO)
        for name in ("epsm_host_core_grad_f32", "epsm_host_core_grad_f64"):
            fn = getattr(_lib, name)
            fn.restype = C.c_int
            fn.argtypes = [C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p,
                           C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_double,
                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    return _lib


def host_core_calc_grad(variant, path_info, dlduv, dldp, clip=0.1, dtype=torch.float32, dlduv_cols=None):
    rec = PackedRecords(path_info, device="cpu", float_dtype=dtype)
    N, K = rec.N, rec.K
    d = dlduv.detach().to("cpu", dtype).reshape(N, -1).contiguous()
    p = dldp.detach().to("cpu", dtype).reshape(N, 3).contiguous()
    if dlduv_cols is None:
        nzc = torch.nonzero((d != 0).any(dim=0)).flatten()
        dlduv_cols = int(nzc.max()) + 1 if nzc.numel() else 0
        dlduv_cols = max(2, dlduv_cols + (dlduv_cols & 1))
    P = num_param_grads(variant, K)
    out_p = torch.full((P, N, 3), float("nan"), dtype=dtype)
    out_l = torch.full((K, N, 3), float("nan"), dtype=dtype)
    out_d = torch.full((K, N, 3), float("nan"), dtype=dtype)
    fn = lib().epsm_host_core_grad_f32 if dtype == torch.float32 else lib().epsm_host_core_grad_f64
    rc = fn(VARIANTS[variant], N, K, rec.cam.data_ptr(), C.addressof(rec.records),
            d.data_ptr(), d.shape[1], dlduv_cols, p.data_ptr(), float(clip),
            out_p.data_ptr(), out_l.data_ptr(), out_d.data_ptr(), 0)
    assert rc == 0
    return list(out_p.unbind(0)), list(out_l.unbind(0)), list(out_d.unbind(0))
