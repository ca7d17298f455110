"""Builds and binds tests/host_harness (the kernel core compiled for the CPU)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import torch

from epsm_mitsuba3_amd.records import PackedRecords, VARIANTS, num_param_grads

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_harness")
_CSRC = os.path.join(os.path.dirname(_DIR), "..", "epsm_mitsuba3_amd", "csrc")
# core name -> (library, source, headers it compiles, entry-point stem)
#   "path": epsm_path_core.h, one lane = one path (the dense calc_grad kernel)
#   "cp":   epsm_cp_core.h, one lane = one (path, constraint vertex) (the fused backward kernel)
CORES = {
    "path": ("libpath_core_host.so", "path_core_host.cpp", ("epsm_path_core.h", "epsm_tangent_core.h"), "epsm_host_core_grad"),
    "cp": ("libcp_core_host.so", "cp_core_host.cpp", ("epsm_path_core.h", "epsm_cp_core.h"), "epsm_host_cp_grad"),
}
_libs = {}


SAN = os.environ.get("EPSM_SAN", "0") == "1"       # tools/run_san.sh: the AddressSanitizer / UBSan builds (Makefile `san`)


def lib(core="path"):
    if core not in _libs:
        so, src, hdrs, stem = CORES[core]
        if SAN:
            so = so.replace(".so", "_san.so")
        from epsm_mitsuba3_amd._lib import build_lock
        with build_lock(_DIR):
            subprocess.run(["make", "-C", _DIR, "-s", os.path.join(_DIR, so)], check=True)      # make decides what is stale
        so = os.path.join(_DIR, so)
        _lib = C.CDLL(so)
        _libs[core] = _lib
        for name in (stem + "_f32", stem + "_f64"):
            fn = getattr(_lib, name)
            fn.restype = C.c_int
            fn.argtypes = [C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p,
                           C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_double,
                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    return _libs[core]


def host_core_calc_grad(variant, path_info, dlduv, dldp, clip=0.1, dtype=torch.float32, dlduv_cols=None, core="path"):
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
    fn = getattr(lib(core), CORES[core][3] + ("_f32" if dtype == torch.float32 else "_f64"))
    rc = fn(VARIANTS[variant], N, K, rec.cam.data_ptr(), C.addressof(rec.records),
            d.data_ptr(), d.shape[1], dlduv_cols, p.data_ptr(), float(clip),
            out_p.data_ptr(), out_l.data_ptr(), out_d.data_ptr(), 0)
    assert rc == 0
    return list(out_p.unbind(0)), list(out_l.unbind(0)), list(out_d.unbind(0))
