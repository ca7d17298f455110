"""Path-vertex records: the reference's ``path_info`` list <-> ``EpsmVertexRecord``.

``path_info`` is what ``EPSMIntegrator.sample_path(log_path=True)`` returns
(epsm.py:547, 648-654): ``[{"cam"}, {vertex 1}, ..., {vertex K}]``.  The C ABI
(include/epsm.h) takes the same arrays by pointer, without copying: (N,3) fp32
row-major vectors, (N) fp32 scalars, (N) u8 masks, (N) u32 BSDF flags.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

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


PACKED_KEYS = ("tri", "aux", "emit", "shadow")


def _loose_to_packed(scatter_info: Sequence[dict], dev) -> tuple:
    """Loose addressing (``vidx (N,3), mode (N), bsdf_id, dhf_dalpha (N,3), evidx (N,3), eb0, eb1, eweight, svidx (N,3),
    sb0, sb1, sdis, smode``: vertex rows per path, the form small tests write by hand) -> packed records + a private
    triangle table holding the distinct ``[v0, v1, v2, mode]`` rows.  A triple with a negative index means "none"."""
    i32 = lambda t: t.detach().to(dev).to(torch.int64)
    bits = lambda t: t.detach().to(dev).to(torch.float32).contiguous().view(torch.int32).to(torch.int64)
    rows, spans = [], []

    def add(vidx, mode):
        v = i32(vidx).reshape(-1, 3)
        m = i32(mode).reshape(-1, 1) if torch.is_tensor(mode) else torch.full((v.shape[0], 1), int(mode), dtype=torch.int64, device=dev)
        spans.append((len(rows), v.shape[0]))
        rows.append(torch.cat([v, m], dim=1))
        return len(rows) - 1
    slots = []
    for rec in scatter_info:
        n = i32(rec["vidx"]).reshape(-1, 3).shape[0]
        slot = {"tri": add(rec["vidx"], rec.get("mode", MODE_POS_ATTACHED | MODE_NRM_ATTACHED | MODE_VERTEX_NORMALS))}
        if rec.get("evidx") is not None:
            slot["emit"] = add(rec["evidx"], MODE_POS_ATTACHED)
        if rec.get("svidx") is not None:
            slot["shadow"] = add(rec["svidx"], rec.get("smode", MODE_POS_ATTACHED))
        slots.append((slot, n))
    allrows = torch.cat(rows, dim=0)
    none = (allrows[:, :3] < 0).any(dim=1) | (allrows[:, :3] >= 0xFFFFFFFF).any(dim=1)
    table, inv = torch.unique(allrows[~none], dim=0, return_inverse=True)
    ids = torch.full((allrows.shape[0],), NO_INDEX, dtype=torch.int64, device=dev)
    ids[~none] = inv
    offs, o = [], 0
    for r in rows:
        offs.append(o); o += r.shape[0]
    tid = lambda j: ids[offs[j]: offs[j] + rows[j].shape[0]]
    out = []
    for (slot, n), rec in zip(slots, scatter_info):
        p = {"tri": tid(slot["tri"]), "aux": None, "emit": None, "shadow": None}
        if rec.get("bsdf_id") is not None and rec.get("dhf_dalpha") is not None:
            p["aux"] = torch.cat([i32(rec["bsdf_id"]).reshape(-1, 1), bits(rec["dhf_dalpha"]).reshape(-1, 3)], dim=1)
        if "emit" in slot:
            p["emit"] = torch.stack([tid(slot["emit"]), bits(rec["eb0"]), bits(rec["eb1"]), bits(rec["eweight"])], dim=1)
        if "shadow" in slot:
            p["shadow"] = torch.stack([tid(slot["shadow"]), bits(rec["sb0"]), bits(rec["sb1"]), bits(rec["sdis"])], dim=1)
        out.append(p)
    if table.shape[0] == 0:
        table = torch.zeros((1, 4), dtype=torch.int64, device=dev)
    return out, table


def loose_from_packed(rec: dict, table: Optional[torch.Tensor] = None) -> dict:
    """The inverse view of ``_loose_to_packed`` for ONE vertex: vertex rows and mode bits per path, looked up in the
    triangle table (``vidx, mode, bsdf_id, dhf_dalpha, evidx, emode, eb0, eb1, eweight, svidx, smode, sb0, sb1, sdis``;
    "no triangle" = rows of -1, mode 0).  What the kernels do with the records, spelled out in torch."""
    table = (rec.get("table") if table is None else table).to(torch.int64)
    T = table.shape[0]

    def rows(ids):
        ids = ids.to(torch.int64) & 0xFFFFFFFF
        ok = ids < T
        r = table[torch.where(ok, ids, torch.zeros_like(ids))]
        # (bits 8.. of the mode word are the alpha slot + 1 of the triangle's BSDF, for the packed log: not a mode bit)
        return torch.where(ok[:, None], r[:, :3], torch.full_like(r[:, :3], -1)), torch.where(ok, r[:, 3] & 0xF, torch.zeros_like(ids))
    f = lambda t: t.contiguous().view(torch.float32)
    out = {}
    out["vidx"], out["mode"] = rows(rec["tri"])
    if rec.get("aux") is not None:
        out["bsdf_id"], out["dhf_dalpha"] = rec["aux"][:, 0], f(rec["aux"][:, 1:4].to(torch.int32))
    if rec.get("emit") is not None:
        e = rec["emit"].to(torch.int32)
        out["evidx"], out["emode"] = rows(e[:, 0])
        out["eb0"], out["eb1"], out["eweight"] = f(e[:, 1]), f(e[:, 2]), f(e[:, 3])
    if rec.get("shadow") is not None:
        h = rec["shadow"].to(torch.int32)
        out["svidx"], out["smode"] = rows(h[:, 0])
        out["sb0"], out["sb1"], out["sdis"] = f(h[:, 1]), f(h[:, 2]), f(h[:, 3])
    return out


class PackedScatter:
    """Per-vertex addressing of the parameter buffers next to ``PackedRecords`` (``EpsmScatterRecord`` + the scene's
    triangle table, include/epsm.h); keeps the tensors alive next to the ctypes array.

    ``scatter_info[k-1]`` holds either the packed arrays -- ``tri (N) int32`` triangle ids, ``aux (N,4)``
    ``[bsdf_id, d hf / d alpha bits]``, ``emit (N,4) [etri, eb0, eb1, eweight bits]``, ``shadow (N,4) [stri, sb0, sb1,
    dis bits]`` (first vertex, ``max_depth <= 3`` only), any of the last three may be ``None`` -- with the table
    ``(T,4) int32 [v0, v1, v2, mode]`` passed as ``table`` or stored under ``scatter_info[0]["table"]``; or the loose
    per-path vertex triples of ``_loose_to_packed``."""

    def __init__(self, scatter_info: Sequence[dict], device, float_dtype=torch.float32, table: Optional[torch.Tensor] = None):
        self.K = len(scatter_info)
        self.device = dev = torch.device(device)
        if self.K and "tri" not in scatter_info[0]:
            packed, table = _loose_to_packed(scatter_info, dev)
        else:
            packed = [{k: rec.get(k) for k in PACKED_KEYS} for rec in scatter_info]
            if table is None and self.K:
                table = scatter_info[0].get("table")
            if table is None:
                raise ValueError("PackedScatter: packed addressing needs the scene's triangle table")
        conv = lambda v: None if v is None else v.detach().to(device=dev, dtype=torch.int32).contiguous()
        self.packed = [{k: conv(v) for k, v in p.items()} for p in packed]
        self.table = conv(table).reshape(-1, 4)
        self.T = int(self.table.shape[0])
        self.records = (EpsmScatterRecord * max(1, self.K))()
        for k, p in enumerate(self.packed):
            r = self.records[k]
            n = p["tri"].shape[0]
            for name, width in (("aux", 4), ("emit", 4), ("shadow", 4)):
                if p[name] is not None and tuple(p[name].shape) != (n, width):
                    raise ValueError(f"scatter record {name!r}: expected shape {(n, width)}, got {tuple(p[name].shape)}")
            if p["tri"].dim() != 1:
                raise ValueError(f"scatter record 'tri': expected {n} triangle ids, got shape {tuple(p['tri'].shape)}")
            r.tri = p["tri"].data_ptr()
            r.aux = p["aux"].data_ptr() if p["aux"] is not None else None
            r.emit = p["emit"].data_ptr() if p["emit"] is not None else None
            r.shadow = p["shadow"].data_ptr() if p.get("shadow") is not None else None

    def table_ptr(self) -> int:
        return self.table.data_ptr()


class EpsmPackedLog(C.Structure):
    """Mirror of ``struct EpsmPackedLog`` (include/epsm.h, ABI v7)."""
    _fields_ = [(n, C.c_void_p) for n in ("rays", "flags", "verts", "shadow")] + [("ray_stride", C.c_int64), ("path_stride", C.c_int64),
                                                                                ("path_list", C.c_void_p), ("path_count", C.c_void_p)]


FLAG_DIFFUSE, FLAG_NULL, FLAG_ACTIVE, FLAG_ACTIVE_EM, FLAG_ISMESH = 1, 2, 4, 8, 16
REC_WORDS = 32
LOG_LAYOUTS = ("interleaved", "dense")
DEFAULT_LOG_LAYOUT = "dense"


def alloc_log(n_paths: int, K: int, device, layout: Optional[str] = None, zero: bool = False):
    """Storage of a native log of ``n_paths`` paths with K records each: ``(rays (N,12), verts (N,K,32))``.

    ``"interleaved"`` (include/epsm.h, EpsmPackedLog): ONE block of K + 1 cache lines per path -- words 0..11 the rays,
    16.. the records, i.e. the records lie half a line off the lines, so that a path's lanes read a run of whole lines
    ([rays | first sector of vertex 1], [second sector of vertex k | first sector of vertex k + 1]); both tensors are
    views of it.  ``"dense"`` (the default): two contiguous arrays.  Measured (MEASUREMENTS.md 10.12): the interleaved block
    cuts the backward kernel's HBM reads by 16 % and its time by 1 % on the headline slab, and costs 8 % on the traced scene
    (the tracer's 48-byte ray stores no longer coalesce) -- so the tracer and the benchmark keep the dense arrays."""
    layout = layout or DEFAULT_LOG_LAYOUT
    if layout not in LOG_LAYOUTS:
        raise ValueError(f"unknown log layout {layout!r}")
    make = torch.zeros if zero else torch.empty
    if layout == "dense":
        return make((n_paths, 12), device=device, dtype=torch.float32), make((n_paths, K, REC_WORDS), device=device, dtype=torch.float32)
    block = make((n_paths, REC_WORDS * (K + 1)), device=device, dtype=torch.float32)
    return block[:, 0:12], block[:, 16:16 + REC_WORDS * K].unflatten(1, (K, REC_WORDS))


class PackedLog:
    """The native path log (``EpsmPackedLog``): rays ``(N,12)``, one flag word per path, ONE 128-byte record per
    (path, vertex); ``rays`` and ``verts`` are either two contiguous arrays or views of one interleaved block per path
    (``alloc_log``).  The tracer writes it directly (``Scene`` with ``packed_log``); ``from_trace`` builds it from the
    per-array records of a ``PathTrace`` (synthetic wavefronts, tests) -- the triangle table must then carry the alpha
    slots in bits 8.. of its mode words, because the packed record has no room for a per-vertex BSDF id."""

    def __init__(self, rays, flags, verts, shadow, table, K):
        self.rays, self.flags, self.verts, self.shadow, self.table, self.K = rays, flags, verts, shadow, table, int(K)
        self.N = int(rays.shape[0])
        self.T = int(table.shape[0])
        self.device = rays.device
        assert tuple(rays.shape) == (self.N, 12) and tuple(flags.shape) == (self.N,) and tuple(verts.shape) == (self.N, self.K, REC_WORDS)
        for t in (flags, table) + ((shadow,) if shadow is not None else ()):
            assert t.is_contiguous() and t.element_size() == 4
        assert rays.dtype == torch.float32 and verts.dtype == torch.float32
        # words between consecutive paths; inside a path the rays / a record / the records are contiguous
        self.ray_stride = int(rays.stride(0)) if self.N > 1 else 12
        self.path_stride = int(verts.stride(0)) if self.N > 1 else REC_WORDS * self.K
        assert rays.stride(1) == 1 and verts.stride(2) == 1 and (self.K == 1 or verts.stride(1) == REC_WORDS)
        assert self.ray_stride % 4 == 0 and self.path_stride % 4 == 0 and self.ray_stride >= 12 and self.path_stride >= REC_WORDS * self.K
        self.layout = "dense" if (self.ray_stride, self.path_stride) == (12, REC_WORDS * self.K) else "interleaved"
        self.c = EpsmPackedLog(rays.data_ptr(), flags.data_ptr(), verts.data_ptr(), shadow.data_ptr() if shadow is not None else None,
                               self.ray_stride, self.path_stride)
        # set by the tracer under EPSM_TRACE_FUSE_FIRST_HIT: the paths without a chain (their flag words are 0) and the sum d / d ray.o
        # are already in the gradient buffers -- the backward kernel is then called without grad_o_sum and gives such paths no lane
        self.first_hit_done = False
        self.path_list = self.path_count = None

    def set_path_list(self, path_list: torch.Tensor, path_count: torch.Tensor) -> None:
        """The paths the backward pass is to look at (EpsmPackedLog.path_list / path_count: the tracer's survivors under
        EPSM_TRACE_FUSE_FIRST_HIT); every other path's flag word is 0.  The kernel reads the count on the device."""
        assert path_list.dtype == torch.int32 and path_count.dtype == torch.int32 and path_list.is_contiguous() and path_count.numel() == 1
        self.path_list, self.path_count = path_list, path_count
        self.c.path_list, self.c.path_count = path_list.data_ptr(), path_count.data_ptr()

    def table_ptr(self) -> int:
        return self.table.data_ptr()

    @staticmethod
    def from_trace(trace, device=None, table: Optional[torch.Tensor] = None, free: bool = False, layout: Optional[str] = None) -> "PackedLog":
        dev = torch.device(device) if device is not None else trace.ray_d.device
        f = lambda t: t.detach().to(dev, torch.float32)
        pi, si = trace.path_info, trace.scatter_info
        K = len(pi) - 1
        N = trace.ray_d.shape[0]
        rays, verts = alloc_log(N, K, dev, layout, zero=True)
        for j, t in enumerate((trace.ray_o, trace.ray_d, trace.ray_dx, trace.ray_dy)):
            rays[:, 3 * j: 3 * j + 3] = f(t)
        flags = torch.zeros(N, dtype=torch.int32, device=dev)
        iview = verts.view(torch.int32)
        shadow = None
        if table is None:
            table = si[0].get("table")
        if table is None:
            raise ValueError("PackedLog.from_trace needs the scene's triangle table")
        table = table.detach().to(dev, torch.int32).contiguous()
        for k in range(1, K + 1):
            r, s = pi[k], si[k - 1]
            b = _flags_tensor(r["bsdf"]).to(dev).to(torch.int64)
            m = lambda t: (t.to(dev) > 0) if t.dtype != torch.bool else t.to(dev)
            w = (((b & 6) != 0).to(torch.int32) * FLAG_DIFFUSE | ((b & 1) != 0).to(torch.int32) * FLAG_NULL |
                 m(r["active"]).to(torch.int32) * FLAG_ACTIVE | m(r["active_em"]).to(torch.int32) * FLAG_ACTIVE_EM |
                 m(r["ismesh"]).to(torch.int32) * FLAG_ISMESH)
            flags |= w << (5 * (k - 1))
            v = verts[:, k - 1]
            iv = iview[:, k - 1]
            tri = s["tri"].to(dev, torch.int32)
            # (word layout: include/epsm.h, EpsmPackedLog -- the first sector holds geometry, barycentrics and triangle id)
            for j in range(3):
                v[:, 3 * j: 3 * j + 3] = f(r["points"][j])
            v[:, 9], v[:, 10] = f(r["uv"][0]), f(r["uv"][1])
            iv[:, 11] = tri
            v[:, 12:15], v[:, 15] = f(r["normals"][0]), f(r["eta"])
            v[:, 16:19], v[:, 19:22] = f(r["normals"][1]), f(r["normals"][2])
            light = f(r["light"])
            v[:, 22:24], v[:, 28] = light[:, 0:2], light[:, 2]
            if s.get("emit") is not None:
                iv[:, 24:28] = s["emit"].to(dev, torch.int32)
            else:
                iv[:, 24] = -1
            if s.get("aux") is not None:
                aux = s["aux"].to(dev, torch.int32)
                iv[:, 29:32] = aux[:, 1:4]
                slot = torch.where(tri >= 0, (table[:, 3][tri.clamp_min(0).long()] >> 8) - 1, torch.full_like(tri, -1))
                if not bool((torch.where(aux[:, 0] >= 0, aux[:, 0], torch.full_like(tri, -1)) == slot).all()):
                    raise ValueError("PackedLog: aux[:,0] (alpha slot per vertex) disagrees with bits 8.. of the triangle table")
            if k == 1 and s.get("shadow") is not None:
                shadow = s["shadow"].to(dev, torch.int32).contiguous()
            if free:
                r.clear(); s.clear()
        return PackedLog(rays, flags.contiguous(), verts, shadow, table, K)
