"""``calc_grad`` of the EPSM integrators on MI355X.

Drop-in for ``ManifoldIntegrator.calc_grad`` (epsm.py:745-946) and
``ManifoldCausticIntegrator.calc_grad`` (epsm.py:952-1200): same arguments,
same return structure (three lists of (N,3) tensors), computed by one HIP
kernel launch (``epsm_manifold_grad``, include/epsm.h) instead of ~10^3 torch
ops per call.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from .records import PackedRecords, VARIANTS, num_param_grads

OUTLIER_CLIP = 0.1   # epsm.py:932-944


def _stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def manifold_grad_packed(variant: str, rec: PackedRecords, dlduv: torch.Tensor, dldp: torch.Tensor,
                         clip: float = OUTLIER_CLIP, dlduv_cols: Optional[int] = None,
                         out: Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = None):
    """Launches the kernel on already packed records.  Returns the three output
    tensors ``(P,N,3)``, ``(K,N,3)``, ``(K,N,3)`` (views of ``out`` if given)."""
    if variant not in VARIANTS:
        raise ValueError(f"unknown integrator variant {variant!r}")
    dev = rec.device
    if dev.type != "cuda":
        raise _lib.EpsmError("the EPSM hot path runs on the GPU only (no CPU fallback); "
                             f"records live on {dev}")
    N, K = rec.N, rec.K
    d = dlduv.detach()
    if d.dtype != torch.float32 or d.device != dev:
        d = d.to(device=dev, dtype=torch.float32)
    d = d.reshape(N, d.numel() // N if N > 0 else d.shape[-1])
    if not d.is_contiguous():
        d = d.contiguous()
    p = dldp.detach()
    if p.dtype != torch.float32 or p.device != dev or not p.is_contiguous():
        p = p.to(device=dev, dtype=torch.float32).contiguous()
    if tuple(p.shape) != (N, 3):
        raise ValueError(f"dldp must have shape ({N}, 3), got {tuple(p.shape)}")
    width = d.shape[1]
    if width < 2:
        raise ValueError("dlduv needs at least the two columns of the first vertex")
    if dlduv_cols is None:
        # render_backward only fills columns 0,1 (epsm.py:256,268-269); anything
        # else costs one device-side check here.
        dlduv_cols = 2
        if width > 2 and bool((d[:, 2:] != 0).any()):
            dlduv_cols = width
    P = num_param_grads(variant, K)
    if out is None:
        out_p = torch.empty((P, N, 3), device=dev, dtype=torch.float32)
        out_l = torch.empty((K, N, 3), device=dev, dtype=torch.float32)
        out_d = torch.empty((K, N, 3), device=dev, dtype=torch.float32)
    else:
        out_p, out_l, out_d = out
        assert tuple(out_p.shape) == (P, N, 3) and tuple(out_l.shape) == (K, N, 3) and tuple(out_d.shape) == (K, N, 3)
        assert all(t.is_contiguous() and t.dtype == torch.float32 and t.device == dev for t in out)
    with torch.cuda.device(dev):
        rc = _lib.lib().epsm_manifold_grad(
            VARIANTS[variant], N, K, rec.cam.data_ptr(), C.addressof(rec.records),
            d.data_ptr(), width, int(dlduv_cols), p.data_ptr(), float(clip),
            out_p.data_ptr(), out_l.data_ptr(), out_d.data_ptr(), _stream_ptr(dev))
    _lib.check(rc, "epsm_manifold_grad")
    return out_p, out_l, out_d


def calc_grad(variant: str, path_info: Sequence[dict], dlduv: torch.Tensor, dldp: torch.Tensor,
              Lt=None, clip: float = OUTLIER_CLIP, dlduv_cols: Optional[int] = None
              ) -> Tuple[List[torch.Tensor], List[torch.Tensor], List[torch.Tensor]]:
    """``(final_param_grad, light_grad, diffuse_grad)`` exactly as the reference
    returns them: ``final_param_grad[5(k-1)+{0,1,2,3,4}]`` = gradient w.r.t.
    ``p0,p1,p2,n,m`` of vertex k; ``Lt`` is accepted and ignored like in the
    reference (epsm.py:745)."""
    dev = path_info[0]["cam"].device
    if dev.type != "cuda":
        raise _lib.EpsmError("calc_grad: path_info must live on the GPU (no CPU fallback)")
    rec = PackedRecords(path_info, device=dev)
    out_p, out_l, out_d = manifold_grad_packed(variant, rec, dlduv, dldp, clip=clip, dlduv_cols=dlduv_cols)
    return list(out_p.unbind(0)), list(out_l.unbind(0)), list(out_d.unbind(0))
