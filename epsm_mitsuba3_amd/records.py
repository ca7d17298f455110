"""Path-vertex records: the reference's ``path_info`` list <-> ``EpsmVertexRecord``.

``path_info`` is what ``EPSMIntegrator.sample_path(log_path=True)`` returns
(epsm.py:547, 648-654): ``[{"cam"}, {vertex 1}, ..., {vertex K}]``.  The C ABI
(include/epsm.h) takes the same arrays by pointer, without copying: (N,3) fp32
row-major vectors, (N) fp32 scalars, (N) u8 masks, (N) u32 BSDF flags.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence

import torch

MAX_VERTICES = 5  # EPSM_MAX_VERTICES; epsm.py:648 logs at most 5 bounces

VARIANTS = {"manifold": 0, "manifold_caustic": 1}


class EpsmVertexRecord(C.Structure):
    """Mirror of ``struct EpsmVertexRecord`` (include/epsm.h)."""
    _fields_ = [(n, C.c_void_p) for n in (
        "p0", "p1", "p2", "n0", "n1", "n2", "b0", "b1", "eta", "hf", "light",
        "bsdf", "active", "active_em", "ismesh")]


def num_param_grads(variant: str, K: int) -> int:
    """5K for ``manifold`` (epsm.py:786-788,815-816), 5K-2 for ``manifold_caustic``
    (epsm.py:1102-1105)."""
    return 5 * K - 2 if VARIANTS[variant] == 1 else 5 * K


def _flags_tensor(x) -> torch.Tensor:
    # the reference stores ``bsdf.flags()`` (a Dr.Jit UInt32, epsm.py:649); the
    # stub harness wraps an integer tensor in an object with attribute ``t``.
    if hasattr(x, "t") and isinstance(x.t, torch.Tensor):
        x = x.t
    if hasattr(x, "torch") and not isinstance(x, torch.Tensor):
        x = x.torch()
    return x


class PackedRecords:
    """Keeps the (possibly converted) tensors alive next to the ctypes array."""

    def __init__(self, path_info: Sequence[dict], device=None, float_dtype=torch.float32):
        if len(path_info) < 2:
            raise ValueError("path_info needs the camera entry and at least one vertex")
        self.K = len(path_info) - 1
        self.float_dtype = float_dtype
        self._keep: List[torch.Tensor] = []
        cam = path_info[0]["cam"]
        self.device = torch.device(device) if device is not None else cam.device
        self.cam = self._f(cam, 3)
        self.N = int(self.cam.shape[0])
        self.records = (EpsmVertexRecord * self.K)()
        for k in range(1, self.K + 1):
            rec = path_info[k]
            r = self.records[k - 1]
            pts, nrm, uv = rec["points"], rec["normals"], rec["uv"]
            r.p0, r.p1, r.p2 = (self._f(pts[j], 3).data_ptr() for j in range(3))
            r.n0, r.n1, r.n2 = (self._f(nrm[j], 3).data_ptr() for j in range(3))
            r.b0, r.b1 = self._f(uv[0], 1).data_ptr(), self._f(uv[1], 1).data_ptr()
            r.eta = self._f(rec["eta"], 1).data_ptr()
            r.hf = self._f(rec["hf"], 3).data_ptr() if rec.get("hf") is not None else None
            r.light = self._f(rec["light"], 3).data_ptr()
            r.bsdf = self._conv(_flags_tensor(rec["bsdf"]), torch.int32).data_ptr()
            r.active = self._mask(rec["active"]).data_ptr()
            r.active_em = self._mask(rec["active_em"]).data_ptr()
            r.ismesh = self._mask(rec["ismesh"]).data_ptr()

    # -- helpers ---------------------------------------------------------
    def _keepalive(self, t: torch.Tensor) -> torch.Tensor:
        self._keep.append(t)
        return t

    def _conv(self, t: torch.Tensor, dtype) -> torch.Tensor:
        t = t.detach()
        if t.dtype != dtype or t.device != self.device or not t.is_contiguous():
            t = t.to(device=self.device, dtype=dtype).contiguous()
        if t.shape[0] != getattr(self, "N", t.shape[0]):
            raise ValueError(f"record array has {t.shape[0]} rows, expected {self.N}")
        return self._keepalive(t)

    def _f(self, t: torch.Tensor, width: int) -> torch.Tensor:
        t = self._conv(t, self.float_dtype)
        want = (t.shape[0], 3) if width == 3 else (t.shape[0],)
        if tuple(t.shape) != want:
            raise ValueError(f"expected shape {want}, got {tuple(t.shape)}")
        return t

    def _mask(self, t: torch.Tensor) -> torch.Tensor:
        t = t.detach()
        if t.dtype == torch.bool:
            t = t.to(self.device).contiguous().view(torch.uint8)
            return self._keepalive(t)
        if t.dtype == torch.uint8:
            return self._conv(t, torch.uint8)
        return self._keepalive((t.to(self.device) > 0).to(torch.uint8).contiguous())


class EpsmScatterRecord(C.Structure):
    """Mirror of ``struct EpsmScatterRecord`` (include/epsm.h)."""
    _fields_ = [(n, C.c_void_p) for n in ("tri", "aux", "emit", "shadow")]


MODE_VERTEX_NORMALS, MODE_FLIP_NORMALS, MODE_POS_ATTACHED, MODE_NRM_ATTACHED = 1, 2, 4, 8
NO_INDEX = 0xFFFFFFFF


def pack_scatter_vertex(rec: dict, device, float_dtype=torch.float32) -> dict:
    """One logged vertex's parameter addressing -> the three packed int32 arrays of
    ``EpsmScatterRecord``: ``tri (N,4) = [v0,v1,v2,mode]``, ``aux (N,4) = [bsdf_id, dhf xyz bits]``,
    ``emit (N,8) = [e0,e1,e2, eb0,eb1,eweight bits, 0,0]``, and (first vertex, ``max_depth <= 3`` only)
    ``shadow (N,8) = [s0,s1,s2, sb0,sb1,dis bits, mode, 0]``.  Accepts either the packed keys or the
    loose ones (``vidx, mode, bsdf_id, dhf_dalpha, evidx, eb0, eb1, eweight, svidx, sb0, sb1, sdis, smode``)."""
    dev = torch.device(device)
    if "tri" in rec:
        out = {"tri": rec["tri"], "aux": rec.get("aux"), "emit": rec.get("emit"), "shadow": rec.get("shadow")}
    else:
        i32 = lambda t: t.detach().to(dev).to(torch.int32)
        bits = lambda t: t.detach().to(dev).to(torch.float32).contiguous().view(torch.int32)
        tri = torch.cat([i32(rec["vidx"]).reshape(-1, 3), i32(rec["mode"]).reshape(-1, 1)], dim=1)
        out = {"tri": tri, "aux": None, "emit": None, "shadow": None}
        if rec.get("svidx") is not None:
            n = tri.shape[0]
            out["shadow"] = torch.cat([i32(rec["svidx"]).reshape(-1, 3), bits(rec["sb0"]).reshape(-1, 1),
                                       bits(rec["sb1"]).reshape(-1, 1), bits(rec["sdis"]).reshape(-1, 1),
                                       i32(rec["smode"]).reshape(-1, 1), torch.zeros((n, 1), dtype=torch.int32, device=dev)], dim=1)
        if rec.get("bsdf_id") is not None and rec.get("dhf_dalpha") is not None:
            out["aux"] = torch.cat([i32(rec["bsdf_id"]).reshape(-1, 1), bits(rec["dhf_dalpha"]).reshape(-1, 3)], dim=1)
        if rec.get("evidx") is not None:
            n = tri.shape[0]
            out["emit"] = torch.cat([i32(rec["evidx"]).reshape(-1, 3), bits(rec["eb0"]).reshape(-1, 1),
                                     bits(rec["eb1"]).reshape(-1, 1), bits(rec["eweight"]).reshape(-1, 1),
                                     torch.zeros((n, 2), dtype=torch.int32, device=dev)], dim=1)
    return {k: (None if v is None else v.detach().to(device=dev, dtype=torch.int32).contiguous()) for k, v in out.items()}


class PackedScatter:
    """Per-vertex addressing of the parameter buffers next to ``PackedRecords``; keeps the packed
    tensors alive next to the ctypes array.  ``scatter_info[k-1]``: see ``pack_scatter_vertex``."""

    def __init__(self, scatter_info: Sequence[dict], device, float_dtype=torch.float32):
        self.K = len(scatter_info)
        self.device = torch.device(device)
        self.packed = [pack_scatter_vertex(rec, self.device, float_dtype) for rec in scatter_info]
        self.records = (EpsmScatterRecord * self.K)()
        for k, p in enumerate(self.packed):
            r = self.records[k]
            r.tri = p["tri"].data_ptr()
            r.aux = p["aux"].data_ptr() if p["aux"] is not None else None
            r.emit = p["emit"].data_ptr() if p["emit"] is not None else None
            r.shadow = p["shadow"].data_ptr() if p.get("shadow") is not None else None
