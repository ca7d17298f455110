"""ctypes binding of oracle/libepsm_oracle.so (built by oracle/Makefile).

TEST INFRASTRUCTURE: the checker, never the thing measured or shipped.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import torch

from epsm_mitsuba3_amd.records import PackedRecords, VARIANTS, num_param_grads

_HERE = os.path.dirname(os.path.abspath(__file__))
# EPSM_SAN=1 (tools/run_san.sh): the AddressSanitizer / UBSan build of the same sources (oracle/Makefile SAN=1)
_SAN = os.environ.get("EPSM_SAN", "0") == "1"
_LIB_PATH = os.path.join(_HERE, "libepsm_oracle_san.so" if _SAN else "libepsm_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    from epsm_mitsuba3_amd._lib import build_lock
    with build_lock(_HERE):             # the two ranks of a gloo test both come here; the link step renames into place
        subprocess.run(["make", "-C", _HERE, "-s"] + (["SAN=1"] if _SAN else []) + (["-B"] if force else []), check=True)    # make decides what is stale
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        for name in ("epsm_oracle_calc_grad_f32", "epsm_oracle_calc_grad_f64"):
            fn = getattr(_lib, name)
            fn.restype = C.c_int
            fn.argtypes = [C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p,
                           C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_double,
                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    return _lib


def oracle_calc_grad(variant: str, path_info, dlduv: torch.Tensor, dldp: torch.Tensor,
                     clip: float = 0.1, dtype=torch.float32, nthreads: int = 0,
                     dlduv_cols: int | None = None):
    """CPU restatement of ``calc_grad`` (epsm.py:745 / 952).

    Returns ``(final_param_grad, light_grad, diffuse_grad, threads_used)`` with
    the three lists holding (N,3) CPU tensors of ``dtype``.
    """
    assert dtype in (torch.float32, torch.float64)
    rec = PackedRecords(path_info, device="cpu", float_dtype=dtype)
    N, K = rec.N, rec.K
    d = dlduv.detach().to("cpu", dtype).reshape(N, -1).contiguous()
    p = dldp.detach().to("cpu", dtype).reshape(N, 3).contiguous()
    cols = d.shape[1] if dlduv_cols is None else int(dlduv_cols)
    P = num_param_grads(variant, K)
    out_p = torch.empty((P, N, 3), dtype=dtype)
    out_l = torch.empty((K, N, 3), dtype=dtype)
    out_d = torch.empty((K, N, 3), dtype=dtype)
    fn = lib().epsm_oracle_calc_grad_f32 if dtype == torch.float32 else lib().epsm_oracle_calc_grad_f64
    rc = fn(VARIANTS[variant], N, K, rec.cam.data_ptr(), C.addressof(rec.records),
            d.data_ptr(), d.shape[1], cols, p.data_ptr(), float(clip),
            out_p.data_ptr(), out_l.data_ptr(), out_d.data_ptr(), int(nthreads))
    if rc < 0:
        raise RuntimeError(f"oracle failed with code {rc}")
    return list(out_p.unbind(0)), list(out_l.unbind(0)), list(out_d.unbind(0)), rc


def oracle_cond(variant: str, path_info, dlduv: torch.Tensor, dldp: torch.Tensor, dlduv_cols: int | None = None) -> torch.Tensor:
    """(N,) float64: per path, the largest 2-norm condition number among the systems ``cur`` (epsm.py:844, 909) whose
    solve the path's outputs use (1 where it uses none; inf where a system is not finite), from the float64 oracle.
    SURVEY.md 8c states the fp32 tolerance for cond_2 < 1e4."""
    rec = PackedRecords(path_info, device="cpu", float_dtype=torch.float64)
    N, K = rec.N, rec.K
    d = dlduv.detach().to("cpu", torch.float64).reshape(N, -1).contiguous()
    p = dldp.detach().to("cpu", torch.float64).reshape(N, 3).contiguous()
    cols = d.shape[1] if dlduv_cols is None else int(dlduv_cols)
    P = num_param_grads(variant, K)
    out = [torch.empty((P, N, 3), dtype=torch.float64), torch.empty((K, N, 3), dtype=torch.float64), torch.empty((K, N, 3), dtype=torch.float64)]
    cond = torch.ones(N, dtype=torch.float64)
    l = lib()
    l.epsm_oracle_set_cond_out_f64.argtypes = [C.c_void_p]
    l.epsm_oracle_set_cond_out_f64.restype = None
    l.epsm_oracle_set_cond_out_f64(cond.data_ptr())
    try:
        rc = l.epsm_oracle_calc_grad_f64(VARIANTS[variant], N, K, rec.cam.data_ptr(), C.addressof(rec.records),
                                         d.data_ptr(), d.shape[1], cols, p.data_ptr(), 0.1,
                                         out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), 0)
    finally:
        l.epsm_oracle_set_cond_out_f64(None)
    if rc < 0:
        raise RuntimeError(f"oracle failed with code {rc}")
    return cond


# ---------------------------------------------------------------------------
# tangent / scatter restatements (oracle/epsm_oracle_aux.c; "parity unpinned")
# ---------------------------------------------------------------------------
def _aux():
    l = lib()
    if not hasattr(l, "_aux_ready"):
        l.epsm_oracle_first_vertex_tangent.restype = C.c_int
        l.epsm_oracle_first_vertex_tangent.argtypes = [
            C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
            C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
            C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        l.epsm_oracle_scatter.restype = C.c_int
        l.epsm_oracle_scatter.argtypes = [
            C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
            C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64]
        l._aux_ready = True
    return l


def oracle_first_vertex_tangent(ray_o, ray_d, ray_dx, ray_dy, grad_in, spp, res, p0, p1, p2, active, width=2,
                                path_offset=0):
    f = lambda t: t.detach().to("cpu", torch.float32).contiguous()
    o, d, dx, dy, g, q0, q1, q2 = map(f, (ray_o, ray_d, ray_dx, ray_dy, grad_in, p0, p1, p2))
    a = (active.detach().cpu() > 0).to(torch.uint8).contiguous()
    N = d.shape[0]
    dlduv = torch.empty((N, 1, width), dtype=torch.float64)
    dldp = torch.empty((N, 3), dtype=torch.float64)
    go = torch.empty(3, dtype=torch.float64)
    rc = _aux().epsm_oracle_first_vertex_tangent(
        N, int(path_offset), int(spp), int(res), o.data_ptr(), d.data_ptr(), dx.data_ptr(), dy.data_ptr(), g.data_ptr(),
        int(g.shape[1]), int(g.shape[2]), q0.data_ptr(), q1.data_ptr(), q2.data_ptr(), a.data_ptr(),
        dlduv.data_ptr(), width, dldp.data_ptr(), go.data_ptr())
    assert rc == 0
    return dlduv, dldp, go


def oracle_intersect_tangent(o, d, odot, ddot, p0, p1, p2):
    """One ray / one triangle in float64 dual numbers (oracle/epsm_oracle_aux.c, mesh.h:349-362 + mesh.cpp:698-709):
    dict(t, u, v, dt, du, dv, dp (3)) when the origin moves with ``odot`` and the direction with ``ddot``."""
    import numpy as np
    l = _aux()
    arr = [np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(3)) for a in (o, d, odot, ddot, p0, p1, p2)]
    out = np.zeros(9)
    l.epsm_oracle_intersect_tangent.restype = C.c_int
    rc = l.epsm_oracle_intersect_tangent(*[a.ctypes.data_as(C.c_void_p) for a in arr], out.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return dict(t=out[0], u=out[1], v=out[2], dt=out[3], du=out[4], dv=out[5], dp=out[6:9].copy())


def oracle_scatter(variant, path_info, scatter_info, out_param, out_light, out_diffuse, V, B):
    """Deterministic float64 accumulation; returns (grad_pos, grad_nrm, grad_alpha)."""
    from epsm_mitsuba3_amd.records import PackedScatter
    rec = PackedRecords(path_info, device="cpu")
    sc = PackedScatter(scatter_info, device="cpu")
    f = lambda t: (torch.stack(list(t)) if isinstance(t, (list, tuple)) else t).detach().to("cpu", torch.float64).contiguous()
    op, ol, od = f(out_param), f(out_light), f(out_diffuse)
    gp = torch.zeros((V, 3), dtype=torch.float64)
    gn = torch.zeros((V, 3), dtype=torch.float64)
    ga = torch.zeros((max(B, 1),), dtype=torch.float64)
    rc = _aux().epsm_oracle_scatter(VARIANTS[variant], rec.N, rec.K, C.addressof(rec.records), C.addressof(sc.records),
                                    sc.table_ptr(), sc.T, op.data_ptr(), ol.data_ptr(), od.data_ptr(),
                                    gp.data_ptr(), gn.data_ptr(), ga.data_ptr(), V, B)
    assert rc == 0
    return gp, gn, ga[:B]


# ---------------------------------------------------------------------------
# brute-force float64 intersector (oracle/epsm_oracle_trace.c; the tracer's non-self oracle)
# ---------------------------------------------------------------------------
def oracle_intersect(o, d, tri_verts, tmin=0.0, tmax=float("inf"), skip=None, any_hit=False):
    """Every ray against every triangle in float64.  ``o, d``: (n,3); ``tri_verts``: (T,3,3) = p0,p1,p2;
    ``tmin`` / ``tmax``: scalars or (n); ``skip``: (n) triangle to ignore per ray (-1 none).
    Returns dict(tri (n) int64 [-1 = miss], t, u, v, second_t (n) float64): mesh.h:343-365's (t, u, v)."""
    l = lib()
    if not hasattr(l, "_trace_ready"):
        l.epsm_oracle_intersect.restype = C.c_int
        l.epsm_oracle_intersect.argtypes = [C.c_int64] + [C.c_void_p] * 4 + [C.c_int64, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 5
        l._trace_ready = True
    f = lambda t: torch.as_tensor(t).detach().to("cpu", torch.float64).contiguous()
    o, d = f(o).reshape(-1, 3), f(d).reshape(-1, 3)
    n = o.shape[0]
    tv = f(tri_verts).reshape(-1, 9)
    lo = f(tmin).expand(n).contiguous() if f(tmin).dim() == 0 else f(tmin)
    hi = f(tmax).expand(n).contiguous() if f(tmax).dim() == 0 else f(tmax)
    sk = None if skip is None else torch.as_tensor(skip).detach().to("cpu", torch.int64).contiguous()
    tri = torch.empty(n, dtype=torch.int64)
    t, u, v, s2 = (torch.empty(n, dtype=torch.float64) for _ in range(4))
    rc = l.epsm_oracle_intersect(n, o.data_ptr(), d.data_ptr(), lo.data_ptr(), hi.data_ptr(), tv.shape[0], tv.data_ptr(),
                                 sk.data_ptr() if sk is not None else None, int(bool(any_hit)),
                                 tri.data_ptr(), t.data_ptr(), u.data_ptr(), v.data_ptr(), s2.data_ptr())
    assert rc == 0
    return {"tri": tri, "t": t, "u": u, "v": v, "second_t": s2}
