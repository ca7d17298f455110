"""The two stages of ``EPSMIntegrator.render_backward`` (epsm.py:84-306) around
``calc_grad``: the first-vertex tangent (epsm.py:238-272) and the scatter of the
per-path gradients into the scene-parameter gradient buffers (epsm.py:283-297)."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib
from .records import PackedLog, PackedRecords, PackedScatter, VARIANTS, num_param_grads


def _f32(t: torch.Tensor, dev, shape=None) -> torch.Tensor:
    t = t.detach()
    if t.dtype != torch.float32 or t.device != dev or not t.is_contiguous():
        t = t.to(device=dev, dtype=torch.float32).contiguous()
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"expected shape {tuple(shape)}, got {tuple(t.shape)}")
    return t


def first_vertex_tangent(ray_o, ray_d, ray_dx, ray_dy, grad_in: torch.Tensor, spp: int, res: int,
                         p0, p1, p2, active, dlduv_width: int = 2, want_origin_grad: bool = False,
                         path_offset: int = 0
                         ) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]:
    """``(dlduv (N,1,width), dldp1 (N,3), grad_o (3,) | None)`` -- epsm.py:238-272.

    ``grad_in`` is the (H,W,5) gradient image; its top-left ``res x res`` crop is used
    (epsm.py:240).  ``width`` = 2 gives the compact tangent buffer the kernels prefer;
    ``2L`` reproduces the reference's zero-padded ``(N,1,2L)`` tensor."""
    dev = ray_d.device
    if dev.type != "cuda":
        raise _lib.EpsmError("first_vertex_tangent: inputs must live on the GPU (no CPU fallback)")
    N = ray_d.shape[0]
    o, d, dx, dy = (_f32(t, dev, (N, 3)) for t in (ray_o, ray_d, ray_dx, ray_dy))
    q0, q1, q2 = (_f32(t, dev, (N, 3)) for t in (p0, p1, p2))
    g = _f32(grad_in, dev)
    if g.dim() != 3 or g.shape[2] < 5 or g.shape[0] < res or g.shape[1] < res:
        raise ValueError(f"grad_in must be (H>=res, W>=res, >=5), got {tuple(g.shape)}")
    a = active.detach()
    a = (a.to(dev).contiguous().view(torch.uint8) if a.dtype == torch.bool
         else (a.to(dev) > 0).to(torch.uint8).contiguous())
    dlduv = torch.empty((N, 1, dlduv_width), device=dev, dtype=torch.float32)
    dldp = torch.empty((N, 3), device=dev, dtype=torch.float32)
    go = torch.zeros(3, device=dev, dtype=torch.float32) if want_origin_grad else None
    with torch.cuda.device(dev):
        rc = _lib.lib().epsm_first_vertex_tangent(
            N, int(path_offset), int(spp), int(res), o.data_ptr(), d.data_ptr(), dx.data_ptr(), dy.data_ptr(),
            g.data_ptr(), int(g.shape[1]), int(g.shape[2]), q0.data_ptr(), q1.data_ptr(), q2.data_ptr(),
            a.data_ptr(), dlduv.data_ptr(), int(dlduv_width), dldp.data_ptr(),
            go.data_ptr() if go is not None else None, torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc, "epsm_first_vertex_tangent")
    return dlduv, dldp, go


def scatter(variant: str, rec: PackedRecords, sc: PackedScatter,
            out_param: torch.Tensor, out_light: torch.Tensor, out_diffuse: torch.Tensor,
            grad_pos: torch.Tensor, grad_nrm: torch.Tensor, grad_alpha: Optional[torch.Tensor] = None) -> None:
    """Accumulates into ``grad_pos (V,3)``, ``grad_nrm (V,3)``, ``grad_alpha (B)`` in place."""
    dev = rec.device
    if dev.type != "cuda":
        raise _lib.EpsmError("scatter: records must live on the GPU (no CPU fallback)")
    N, K = rec.N, rec.K
    P = num_param_grads(variant, K)
    assert tuple(out_param.shape) == (P, N, 3) and tuple(out_light.shape) == (K, N, 3) and tuple(out_diffuse.shape) == (K, N, 3)
    for t in (out_param, out_light, out_diffuse, grad_pos, grad_nrm):
        assert t.is_contiguous() and t.dtype == torch.float32 and t.device == dev
    V = grad_pos.shape[0]
    assert tuple(grad_pos.shape) == (V, 3) and tuple(grad_nrm.shape) == (V, 3)
    B = 0
    if grad_alpha is not None:
        assert grad_alpha.is_contiguous() and grad_alpha.dtype == torch.float32 and grad_alpha.device == dev
        B = grad_alpha.numel()
    with torch.cuda.device(dev):
        rc = _lib.lib().epsm_scatter(
            VARIANTS[variant], N, K, C.addressof(rec.records), C.addressof(sc.records), sc.table_ptr(), sc.T,
            out_param.data_ptr(), out_light.data_ptr(), out_diffuse.data_ptr(),
            grad_pos.data_ptr(), grad_nrm.data_ptr(), grad_alpha.data_ptr() if grad_alpha is not None else None,
            V, B, torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc, "epsm_scatter")


def manifold_grad_scatter(variant: str, rec: PackedRecords, sc: PackedScatter, dlduv: torch.Tensor,
                          dldp: torch.Tensor, grad_pos: torch.Tensor, grad_nrm: torch.Tensor,
                          grad_alpha: Optional[torch.Tensor] = None, clip: float = 0.1,
                          dlduv_cols: int = 2) -> None:
    """calc_grad + scatter fused in one launch (``epsm_manifold_grad_scatter``): the per-path
    gradient lists are never written; results accumulate into the buffers in place."""
    dev = rec.device
    if dev.type != "cuda":
        raise _lib.EpsmError("manifold_grad_scatter: records must live on the GPU (no CPU fallback)")
    N, K = rec.N, rec.K
    d = _f32(dlduv, dev).reshape(N, dlduv.numel() // N if N else dlduv.shape[-1])
    p = _f32(dldp, dev, (N, 3))
    for t in (grad_pos, grad_nrm):
        assert t.is_contiguous() and t.dtype == torch.float32 and t.device == dev
    V = grad_pos.shape[0]
    assert tuple(grad_pos.shape) == (V, 3) and tuple(grad_nrm.shape) == (V, 3)
    B = 0
    if grad_alpha is not None:
        assert grad_alpha.is_contiguous() and grad_alpha.dtype == torch.float32 and grad_alpha.device == dev
        B = grad_alpha.numel()
    with torch.cuda.device(dev):
        rc = _lib.lib().epsm_manifold_grad_scatter(
            VARIANTS[variant], N, K, rec.cam.data_ptr(), C.addressof(rec.records), C.addressof(sc.records),
            sc.table_ptr(), sc.T, d.data_ptr(), d.shape[1], int(dlduv_cols), p.data_ptr(), float(clip),
            grad_pos.data_ptr(), grad_nrm.data_ptr(), grad_alpha.data_ptr() if grad_alpha is not None else None,
            V, B, torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc, "epsm_manifold_grad_scatter")


def backward_pass(variant: str, rec: PackedRecords, sc: PackedScatter, ray_o, ray_d, ray_dx, ray_dy,
                  grad_in: torch.Tensor, spp: int, res: int, grad_pos: torch.Tensor, grad_nrm: torch.Tensor,
                  grad_alpha: Optional[torch.Tensor] = None, grad_origin: Optional[torch.Tensor] = None,
                  clip: float = 0.1, path_offset: int = 0) -> None:
    """Tangent + calc_grad + scatter in ONE launch (``epsm_backward_pass``): everything ``render_backward`` does
    between the trace and the parameter gradients (epsm.py:238-297).  ``grad_origin`` (3,) accumulates the
    camera-origin gradient; all results accumulate in place."""
    dev = rec.device
    if dev.type != "cuda":
        raise _lib.EpsmError("backward_pass: records must live on the GPU (no CPU fallback)")
    N, K = rec.N, rec.K
    o, d, dx, dy = (_f32(t, dev, (N, 3)) for t in (ray_o, ray_d, ray_dx, ray_dy))
    g = _f32(grad_in, dev)
    if g.dim() != 3 or g.shape[2] < 5 or g.shape[0] < res or g.shape[1] < res:
        raise ValueError(f"grad_in must be (H>=res, W>=res, >=5), got {tuple(g.shape)}")
    for t in (grad_pos, grad_nrm):
        assert t.is_contiguous() and t.dtype == torch.float32 and t.device == dev
    V = grad_pos.shape[0]
    assert tuple(grad_pos.shape) == (V, 3) and tuple(grad_nrm.shape) == (V, 3)
    B = 0
    if grad_alpha is not None:
        assert grad_alpha.is_contiguous() and grad_alpha.dtype == torch.float32 and grad_alpha.device == dev
        B = grad_alpha.numel()
    if grad_origin is not None:
        assert grad_origin.is_contiguous() and grad_origin.dtype == torch.float32 and grad_origin.numel() == 3
    with torch.cuda.device(dev):
        rc = _lib.lib().epsm_backward_pass(
            VARIANTS[variant], N, K, int(path_offset), int(spp), int(res), o.data_ptr(), d.data_ptr(), dx.data_ptr(),
            dy.data_ptr(), g.data_ptr(), int(g.shape[1]), int(g.shape[2]), C.addressof(rec.records), C.addressof(sc.records),
            sc.table_ptr(), sc.T, float(clip), grad_pos.data_ptr(), grad_nrm.data_ptr(),
            grad_alpha.data_ptr() if grad_alpha is not None else None,
            grad_origin.data_ptr() if grad_origin is not None else None, V, B,
            torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc, "epsm_backward_pass")


def backward_pass_packed(variant: str, log: PackedLog, grad_in: torch.Tensor, spp: int, res: int,
                         grad_pos: torch.Tensor, grad_nrm: torch.Tensor, grad_alpha: Optional[torch.Tensor] = None,
                         grad_origin: Optional[torch.Tensor] = None, clip: float = 0.1, path_offset: int = 0) -> None:
    """``backward_pass`` on the native log (``epsm_backward_pass_packed``): one 128-byte record per (path, vertex)."""
    dev = log.device
    if dev.type != "cuda":
        raise _lib.EpsmError("backward_pass_packed: the log must live on the GPU (no CPU fallback)")
    g = _f32(grad_in, dev)
    if g.dim() != 3 or g.shape[2] < 5 or g.shape[0] < res or g.shape[1] < res:
        raise ValueError(f"grad_in must be (H>=res, W>=res, >=5), got {tuple(g.shape)}")
    for t in (grad_pos, grad_nrm):
        assert t.is_contiguous() and t.dtype == torch.float32 and t.device == dev
    V = grad_pos.shape[0]
    assert tuple(grad_pos.shape) == (V, 3) and tuple(grad_nrm.shape) == (V, 3)
    B = 0
    if grad_alpha is not None:
        assert grad_alpha.is_contiguous() and grad_alpha.dtype == torch.float32 and grad_alpha.device == dev
        B = grad_alpha.numel()
    if grad_origin is not None:
        assert grad_origin.is_contiguous() and grad_origin.dtype == torch.float32 and grad_origin.numel() == 3
    with torch.cuda.device(dev):
        rc = _lib.lib().epsm_backward_pass_packed(
            VARIANTS[variant], log.N, log.K, int(path_offset), int(spp), int(res), C.addressof(log.c), g.data_ptr(),
            int(g.shape[1]), int(g.shape[2]), log.table_ptr(), log.T, float(clip), grad_pos.data_ptr(), grad_nrm.data_ptr(),
            grad_alpha.data_ptr() if grad_alpha is not None else None,
            grad_origin.data_ptr() if grad_origin is not None else None, V, B, torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc, "epsm_backward_pass_packed")
