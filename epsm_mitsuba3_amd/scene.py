"""Scene description for the native tracer: the subset of ``mi.load_dict`` that the EPSM
experiments use (EPSM/exp/*.py): ``obj`` / inline meshes / ``rectangle``, ``diffuse``,
``conductor``, ``roughconductor``, ``dielectric``, ``twosided``, ``area`` / ``point`` / ``constant`` / ``envmap`` emitters,
``perspective`` sensors with ``hdrfilm`` (+ ``box`` / ``gaussian`` rfilter) and an
``independent`` sampler.  Geometry is flattened into the arrays of ``EpsmScene``
(include/epsm_trace.h); a median-split BVH is built on the host with numpy.

    scene = Scene.from_dict({...})                       # mi.load_dict shape
    scene.attach("glass", positions=True)                # dr.enable_grad(params['glass.vertex_positions'])
    params = scene.param_grads()                          # ParamGrads over the concatenated vertex buffers
    img = integrator.render(scene, sensor=1, seed=0, spp=16)
    integrator.render_backward(scene, params, grad_in, seed=0)
    scene.set_vertex_positions("glass", new_positions)    # params.update()
"""
from __future__ import annotations

import ctypes as C
import hashlib
import math
import os
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from . import dist as _dist
from .params import ParamGrads

MESH_VERTEX_NORMALS, MESH_FLIP_NORMALS, MESH_POS_ATTACHED, MESH_NRM_ATTACHED, MESH_IS_MESH, MESH_HAS_UV = 1, 2, 4, 8, 16, 32
BSDF_TYPES = {"diffuse": 0, "conductor": 1, "roughconductor": 2, "dielectric": 3}

# complex IORs at R,G,B for the `material` names the experiments use (approximate: Mitsuba
# integrates measured spectra, data files that are not in the reference tree)
CONDUCTORS = {
    "none": ((0.0, 0.0, 0.0), (1.0, 1.0, 1.0)),
    "Al": ((1.657460, 0.880369, 0.521229), (9.223869, 6.269523, 4.837001)),
    "Cu": ((0.200438, 0.924033, 1.102212), (3.912949, 2.452848, 2.142188)),
    "Au": ((0.143119, 0.374957, 1.442479), (3.983160, 2.385721, 1.603215)),
    "Ag": ((0.155265, 0.116723, 0.138342), (4.828427, 3.122499, 2.147038)),
}
IOR = {"vacuum": 1.0, "air": 1.000277, "water": 1.3330, "bk7": 1.5046, "glass": 1.5046, "diamond": 2.419}


# ---------------------------------------------------------------------------- ctypes mirrors
EPSM_TRACE_SPARSE_LOG = 0x1          # include/epsm_trace.h
EPSM_TRACE_PACKED_LOG = 0x2
EPSM_TRACE_GRADIENT_ONLY = 0x4
EPSM_TRACE_GRADIENT_CAUSTIC = 0x8
EPSM_TRACE_NO_TAIL = 0x10
EPSM_TRACE_FUSE_FIRST_HIT = 0x20


class EpsmMesh(C.Structure):
    _fields_ = [("tri_begin", C.c_uint32), ("tri_count", C.c_uint32), ("flags", C.c_uint32), ("bsdf", C.c_int32),
                ("emitter", C.c_int32), ("area", C.c_float), ("cdf_begin", C.c_uint32), ("pad", C.c_uint32)]


class EpsmBsdf(C.Structure):
    _fields_ = [("type", C.c_uint32), ("twosided", C.c_uint32), ("distr", C.c_uint32), ("sample_visible", C.c_uint32),
                ("reflectance", C.c_float * 3), ("alpha", C.c_float), ("eta", C.c_float * 3), ("k", C.c_float * 3),
                ("int_ior", C.c_float), ("ext_ior", C.c_float), ("alpha_slot", C.c_int32), ("color_slot", C.c_int32),
                ("texture", C.c_int32), ("pad", C.c_uint32)]


class EpsmTexture(C.Structure):
    _fields_ = [("texels", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32), ("nearest", C.c_uint32), ("pad", C.c_uint32)]


class EpsmEmitter(C.Structure):
    _fields_ = [("type", C.c_uint32), ("mesh", C.c_int32), ("radiance", C.c_float * 3), ("position", C.c_float * 3),
                ("color_slot", C.c_int32), ("pad", C.c_uint32)]


class EpsmSensor(C.Structure):
    _fields_ = [("to_world", C.c_float * 12), ("sample_to_camera", C.c_float * 16), ("dx", C.c_float * 3),
                ("dy", C.c_float * 3), ("near_clip", C.c_float), ("far_clip", C.c_float),
                ("width", C.c_int32), ("height", C.c_int32), ("border", C.c_int32), ("pad", C.c_int32)]


class EpsmEnvironment(C.Structure):
    _fields_ = [("kind", C.c_int32), ("emitter", C.c_int32), ("width", C.c_int32), ("height", C.c_int32), ("texels", C.c_void_p), ("row_cdf", C.c_void_p),
                ("col_cdf", C.c_void_p), ("cell_pdf", C.c_void_p), ("to_local", C.c_float * 9), ("center", C.c_float * 3),
                ("radius", C.c_float)]


class EpsmSceneC(C.Structure):
    _fields_ = [("positions", C.c_void_p), ("normals", C.c_void_p), ("tri", C.c_void_p), ("tri_mesh", C.c_void_p),
                ("meshes", C.c_void_p), ("n_meshes", C.c_int32), ("bsdfs", C.c_void_p), ("n_bsdfs", C.c_int32),
                ("emitters", C.c_void_p), ("n_emitters", C.c_int32), ("emitter_cdf", C.c_void_p),
                ("bvh", C.c_void_p), ("n_nodes", C.c_int32), ("prim_index", C.c_void_p), ("tri_verts", C.c_void_p),
                ("n_vertices", C.c_int64), ("n_triangles", C.c_int64), ("env", EpsmEnvironment),
                ("texcoords", C.c_void_p), ("textures", C.c_void_p), ("n_textures", C.c_int32)]


class EpsmFirstHitBackward(C.Structure):
    """Mirror of ``struct EpsmFirstHitBackward`` (include/epsm_trace.h)."""
    _fields_ = [("grad_img", C.c_void_p), ("img_width", C.c_int), ("img_channels", C.c_int), ("res", C.c_int), ("clip", C.c_float),
                ("tri_table", C.c_void_p), ("T", C.c_int64), ("V", C.c_int64), ("grad_pos", C.c_void_p), ("grad_o_sum", C.c_void_p),
                ("survivors", C.c_void_p), ("survivor_count", C.c_void_p)]


class EpsmRecordOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "p0", "p1", "p2", "p", "n0", "n1", "n2", "normal", "b0", "b1", "eta", "hf", "light",
        "bsdf", "active", "active_em", "ismesh", "tri", "aux", "emit", "packed", "pflags", "shadow")] + [
        ("ray_stride", C.c_int64), ("packed_stride", C.c_int64), ("first_hit", C.c_void_p)]


# ---------------------------------------------------------------------------- transforms
def look_at(origin, target, up) -> np.ndarray:
    """Transform4f.look_at: camera space +z is the viewing direction."""
    o, t, u = (np.asarray(v, dtype=np.float64) for v in (origin, target, up))
    d = (t - o) / np.linalg.norm(t - o)
    left = np.cross(u, d); left /= np.linalg.norm(left)
    new_up = np.cross(d, left)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = left, new_up, d, o
    return m


def translate(v) -> np.ndarray:
    m = np.eye(4); m[:3, 3] = v; return m


def scale(v) -> np.ndarray:
    v = np.broadcast_to(np.asarray(v, dtype=np.float64), (3,))
    return np.diag([v[0], v[1], v[2], 1.0])


def rotate(axis, angle_deg) -> np.ndarray:
    a = np.asarray(axis, dtype=np.float64); a = a / np.linalg.norm(a)
    th = math.radians(angle_deg); c, s = math.cos(th), math.sin(th)
    x, y, z = a
    r = np.array([[c + x * x * (1 - c), x * y * (1 - c) - z * s, x * z * (1 - c) + y * s],
                  [y * x * (1 - c) + z * s, c + y * y * (1 - c), y * z * (1 - c) - x * s],
                  [z * x * (1 - c) - y * s, z * y * (1 - c) + x * s, c + z * z * (1 - c)]])
    m = np.eye(4); m[:3, :3] = r; return m


def perspective_projection(width: int, height: int, fov_x_deg: float, near: float, far: float, crop=None) -> np.ndarray:
    """camera_to_sample (include/mitsuba/render/sensor.h:227-262 perspective_projection): ``width`` x ``height`` = the FULL film,
    ``crop`` = (crop_width, crop_height, crop_offset_x, crop_offset_y) or None -- the crop window is what maps to [0,1]^2."""
    aspect = width / height
    recip = 1.0 / (far - near)
    cot = 1.0 / math.tan(math.radians(fov_x_deg * 0.5))
    persp = np.array([[cot, 0, 0, 0], [0, cot, 0, 0], [0, 0, far * recip, -near * far * recip], [0, 0, 1, 0]], dtype=np.float64)
    m = scale([-0.5, -0.5 * aspect, 1.0]) @ translate([-1.0, -1.0 / aspect, 0.0]) @ persp
    if crop is not None:
        cw, ch, ox, oy = crop
        m = scale([width / cw, height / ch, 1.0]) @ translate([-ox / width, -oy / height, 0.0]) @ m
    return m


def _xform_point(m, p):
    q = m @ np.array([p[0], p[1], p[2], 1.0])
    return q[:3] / q[3]


# ---------------------------------------------------------------------------- meshes
def load_obj(path: str, with_uv: bool = False):
    """Minimal Wavefront OBJ reader: v / vn / f (triangulated fans), vertices merged per (v, vn) pair.  ``with_uv``: also
    vt, vertices merged per (v, vt, vn) triple, v flipped as obj.cpp:267 does -- returns (v, n, f, uv)."""
    if with_uv:
        return _load_obj_uv(path)
    vs, vns, key_to_idx, out_v, out_n, faces = [], [], {}, [], [], []
    with open(path) as f:
        for line in f:
            t = line.split()
            if not t:
                continue
            if t[0] == "v":
                vs.append([float(x) for x in t[1:4]])
            elif t[0] == "vn":
                vns.append([float(x) for x in t[1:4]])
            elif t[0] == "f":
                idx = []
                for tok in t[1:]:
                    parts = tok.split("/")
                    vi = int(parts[0]); vi = vi - 1 if vi > 0 else len(vs) + vi
                    ni = None
                    if len(parts) >= 3 and parts[2]:
                        ni = int(parts[2]); ni = ni - 1 if ni > 0 else len(vns) + ni
                    key = (vi, ni)
                    if key not in key_to_idx:
                        key_to_idx[key] = len(out_v)
                        out_v.append(vs[vi]); out_n.append(vns[ni] if ni is not None else [0.0, 0.0, 0.0])
                    idx.append(key_to_idx[key])
                for j in range(1, len(idx) - 1):
                    faces.append([idx[0], idx[j], idx[j + 1]])
    v = np.asarray(out_v, dtype=np.float64).reshape(-1, 3)
    n = np.asarray(out_n, dtype=np.float64).reshape(-1, 3)
    has_n = len(vns) > 0
    return v, (n if has_n else None), np.asarray(faces, dtype=np.int64).reshape(-1, 3)


def _load_obj_uv(path: str):
    vs, vts, vns, key_to_idx, out_v, out_t, out_n, faces = [], [], [], {}, [], [], [], []
    with open(path) as f:
        for line in f:
            t = line.split()
            if not t:
                continue
            if t[0] == "v":
                vs.append([float(x) for x in t[1:4]])
            elif t[0] == "vt":
                vts.append([float(t[1]), 1.0 - float(t[2])])
            elif t[0] == "vn":
                vns.append([float(x) for x in t[1:4]])
            elif t[0] == "f":
                idx = []
                for tok in t[1:]:
                    parts = tok.split("/") + ["", ""]
                    rel = lambda x, n: (int(x) - 1 if int(x) > 0 else n + int(x)) if x else None
                    key = (rel(parts[0], len(vs)), rel(parts[1], len(vts)), rel(parts[2], len(vns)))
                    if key not in key_to_idx:
                        key_to_idx[key] = len(out_v)
                        out_v.append(vs[key[0]]); out_t.append(vts[key[1]] if key[1] is not None else [0.0, 0.0])
                        out_n.append(vns[key[2]] if key[2] is not None else [0.0, 0.0, 0.0])
                    idx.append(key_to_idx[key])
                for j in range(1, len(idx) - 1):
                    faces.append([idx[0], idx[j], idx[j + 1]])
    n = np.asarray(out_n, dtype=np.float64).reshape(-1, 3)
    return (np.asarray(out_v, dtype=np.float64).reshape(-1, 3), (n if vns else None), np.asarray(faces, dtype=np.int64).reshape(-1, 3),
            (np.asarray(out_t, dtype=np.float64).reshape(-1, 2) if vts else None))


_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2",
              "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}


def load_ply(path: str):
    """Minimal PLY reader (src/shapes/ply.cpp reads the same subset for the experiment files, e.g. EPSM/exp/glassslab.py:150):
    ascii / binary_little_endian / binary_big_endian; element ``vertex`` with x y z (and nx ny nz), element ``face`` with one list
    property (``vertex_indices`` / ``vertex_index``), polygons triangulated as fans.  Returns (v (V,3), n (V,3) | None, f (F,3))."""
    with open(path, "rb") as fh:
        if fh.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, elements = None, []
        while True:
            line = fh.readline()
            if not line:                                           # end of file inside the header (this looped forever)
                raise ValueError(f"{path}: truncated PLY header (no end_header)")
            t = line.decode("ascii", "replace").split()
            if not t or t[0] == "comment" or t[0] == "obj_info":
                continue
            if t[0] == "format":
                fmt = t[1]
            elif t[0] == "element":
                elements.append({"name": t[1], "count": int(t[2]), "props": []})
            elif t[0] == "property":
                elements[-1]["props"].append(("list", t[2], t[3], t[4]) if t[1] == "list" else ("scalar", t[1], t[2]))
            elif t[0] == "end_header":
                break
        if fmt not in ("ascii", "binary_little_endian", "binary_big_endian"):
            raise ValueError(f"{path}: unsupported PLY format {fmt!r}")
        end = "<" if fmt != "binary_big_endian" else ">"
        verts = faces = None
        tokens = fh.read().split() if fmt == "ascii" else None
        pos = 0
        for el in elements:
            scalars_only = all(p_[0] == "scalar" for p_ in el["props"])
            if scalars_only:
                names = [p_[2] for p_ in el["props"]]
                if fmt == "ascii":
                    k = len(names)
                    arr = np.asarray(tokens[pos:pos + el["count"] * k], dtype=np.float64).reshape(el["count"], k)
                    pos += el["count"] * k
                    table = {nm: arr[:, j] for j, nm in enumerate(names)}
                else:
                    dt = np.dtype([(nm, end + _PLY_TYPES[p_[1]]) for p_, nm in zip(el["props"], names)])
                    rec = np.frombuffer(fh.read(dt.itemsize * el["count"]), dtype=dt, count=el["count"])
                    table = {nm: rec[nm].astype(np.float64) for nm in names}
                if el["name"] == "vertex":
                    verts = table
            else:
                if len(el["props"]) != 1:
                    raise ValueError(f"{path}: element {el['name']} mixes list and scalar properties (unsupported)")
                _, ct, it, _nm = el["props"][0]
                tris = []
                if fmt != "ascii" and el["count"] > 0:
                    # all-triangle binary faces (the common case): ONE read instead of two per face
                    rec = np.dtype([("n", end + _PLY_TYPES[ct]), ("i", end + _PLY_TYPES[it], (3,))])
                    here = fh.tell()
                    buf = fh.read(rec.itemsize * el["count"])
                    a = np.frombuffer(buf, dtype=rec, count=el["count"]) if len(buf) == rec.itemsize * el["count"] else None
                    if a is not None and bool((a["n"] == 3).all()):
                        if el["name"] == "face":
                            faces = a["i"].astype(np.int64).reshape(-1, 3)
                        continue
                    fh.seek(here)                                  # polygons of other sizes: face by face below
                for _ in range(el["count"]):
                    if fmt == "ascii":
                        n_ = int(tokens[pos]); idx = [int(x) for x in tokens[pos + 1:pos + 1 + n_]]; pos += 1 + n_
                    else:
                        n_ = int(np.frombuffer(fh.read(np.dtype(_PLY_TYPES[ct]).itemsize), dtype=end + _PLY_TYPES[ct])[0])
                        idx = np.frombuffer(fh.read(np.dtype(_PLY_TYPES[it]).itemsize * n_), dtype=end + _PLY_TYPES[it]).tolist()
                    for j in range(1, n_ - 1):
                        tris.append([idx[0], idx[j], idx[j + 1]])
                if el["name"] == "face":
                    faces = np.asarray(tris, dtype=np.int64).reshape(-1, 3)
    if verts is None or faces is None or not all(k in verts for k in ("x", "y", "z")):
        raise ValueError(f"{path}: needs a vertex element with x y z and a face element")
    v = np.stack([verts["x"], verts["y"], verts["z"]], axis=1)
    n = np.stack([verts["nx"], verts["ny"], verts["nz"]], axis=1) if all(k in verts for k in ("nx", "ny", "nz")) else None
    return v, n, faces


def vertex_normals(v: np.ndarray, f: np.ndarray) -> np.ndarray:
    """Angle-weighted vertex normals (Mesh::recompute_vertex_normals, src/render/mesh.cpp)."""
    n = np.zeros_like(v)
    p = v[f]
    for i in range(3):
        d0 = p[:, (i + 1) % 3] - p[:, i]; d1 = p[:, (i + 2) % 3] - p[:, i]
        fn = np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0])
        ln = np.linalg.norm(fn, axis=1, keepdims=True)
        fn = np.where(ln > 0, fn / np.maximum(ln, 1e-30), 0)
        cosang = np.sum(d0 * d1, 1) / np.maximum(np.linalg.norm(d0, axis=1) * np.linalg.norm(d1, axis=1), 1e-30)
        ang = np.arccos(np.clip(cosang, -1, 1))
        np.add.at(n, f[:, i], fn * ang[:, None])
    ln = np.linalg.norm(n, axis=1, keepdims=True)
    return np.where(ln > 0, n / np.maximum(ln, 1e-30), np.array([0.0, 0.0, 1.0]))


def vertex_normals_torch(v: torch.Tensor, f: torch.Tensor) -> torch.Tensor:
    """``vertex_normals`` on the device (same angle weighting), for ``Scene.set_vertex_positions``."""
    p = v[f]                                                      # (T,3,3)
    fn = torch.linalg.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0])
    fn = fn / fn.norm(dim=1, keepdim=True).clamp_min(1e-30)
    n = torch.zeros_like(v)
    for i in range(3):
        d0 = p[:, (i + 1) % 3] - p[:, i]; d1 = p[:, (i + 2) % 3] - p[:, i]
        cosang = (d0 * d1).sum(1) / (d0.norm(dim=1) * d1.norm(dim=1)).clamp_min(1e-30)
        n.index_add_(0, f[:, i], fn * torch.acos(cosang.clamp(-1, 1))[:, None])
    ln = n.norm(dim=1, keepdim=True)
    up = torch.tensor([0.0, 0.0, 1.0], device=v.device, dtype=v.dtype)
    return torch.where(ln > 0, n / ln.clamp_min(1e-30), up)


class Mesh:
    def __init__(self, name, v, f, n=None, bsdf=0, emitter=-1, flip_normals=False, is_mesh=True, face_normals=False, uv=None):
        self.name = name
        self.uv = None if uv is None else np.asarray(uv, dtype=np.float64).reshape(-1, 2)
        self.v = np.asarray(v, dtype=np.float64).reshape(-1, 3)
        self.f = np.asarray(f, dtype=np.int64).reshape(-1, 3)
        self.face_normals = face_normals
        self.has_normals = (not face_normals)
        self.n = None
        if self.has_normals:
            self.n = np.asarray(n, dtype=np.float64).reshape(-1, 3) if n is not None else vertex_normals(self.v, self.f)
        self.bsdf, self.emitter = bsdf, emitter
        self.flip_normals, self.is_mesh = flip_normals, is_mesh
        self.pos_attached = self.nrm_attached = False

    def flags(self) -> int:
        return ((MESH_VERTEX_NORMALS if self.has_normals else 0) | (MESH_FLIP_NORMALS if self.flip_normals else 0) |
                (MESH_POS_ATTACHED if self.pos_attached else 0) | (MESH_NRM_ATTACHED if self.nrm_attached else 0) |
                (MESH_IS_MESH if self.is_mesh else 0) | (MESH_HAS_UV if self.uv is not None else 0))


# ---------------------------------------------------------------------------- BVH
# Leaves of <= 6 triangles with the SAH split used all the way down: -12 % trace time on the 128 k-triangle scene
# against leaves of 4 under a median split below 24 triangles (both tracer forms; 2x the build time, paid once:
# vertex updates refit).  Measured 3..7 for both numbers: tools/gpu_bvh_ab.sh.
LEAF_SIZE = 6
SAH_MIN = 6


def _sah_split(cen: np.ndarray, lo: np.ndarray, hi: np.ndarray, bins: int = 16):
    """Best binned surface-area-heuristic plane over the three axes for the triangles (centroids ``cen``, boxes
    ``lo``/``hi``) of one node.  Returns the boolean mask of the left side or None (degenerate extent)."""
    n = cen.shape[0]
    cmin, cmax = cen.min(axis=0), cen.max(axis=0)
    best = (np.inf, None)
    for ax in range(3):
        ext = cmax[ax] - cmin[ax]
        if not ext > 0:
            continue
        b = np.minimum(((cen[:, ax] - cmin[ax]) * (bins / ext)).astype(np.int64), bins - 1)
        cnt = np.bincount(b, minlength=bins)
        blo = np.full((bins, 3), np.inf); bhi = np.full((bins, 3), -np.inf)
        order = np.argsort(b, kind="stable")
        starts = np.searchsorted(b[order], np.arange(bins))
        nz = cnt > 0
        blo[nz] = np.minimum.reduceat(lo[order], starts[nz], axis=0)
        bhi[nz] = np.maximum.reduceat(hi[order], starts[nz], axis=0)
        llo, lhi = np.minimum.accumulate(blo, axis=0), np.maximum.accumulate(bhi, axis=0)
        rlo, rhi = np.minimum.accumulate(blo[::-1], axis=0)[::-1], np.maximum.accumulate(bhi[::-1], axis=0)[::-1]
        nl = np.cumsum(cnt)

        def area(l, h):
            d = np.maximum(h - l, 0.0)
            return d[:, 0] * d[:, 1] + d[:, 1] * d[:, 2] + d[:, 2] * d[:, 0]
        # plane k separates bins [0..k] from [k+1..]
        with np.errstate(invalid="ignore"):
            cost = area(llo[:-1], lhi[:-1]) * nl[:-1] + area(rlo[1:], rhi[1:]) * (n - nl[:-1])
        cost = np.where((nl[:-1] > 0) & (nl[:-1] < n), cost, np.inf)
        k = int(np.argmin(cost))
        if cost[k] < best[0]:
            best = (cost[k], b <= k)
    return best[1]


kMaxWideDepth = 16                      # four-wide levels the traversal stack holds (csrc/epsm_trace_core.h: kBvhStack = 3 x 16)
kMaxBinaryHeight = 2 * kMaxWideDepth    # binary height the collapse folds into them


def build_bvh(pos: np.ndarray, tri: np.ndarray, leaf_size: int = LEAF_SIZE, sah_min: int = SAH_MIN):
    """Binned-SAH BVH (16 bins per axis; nodes of <= ``sah_min`` triangles: median split), built as a binary tree and
    emitted as FOUR-wide nodes (``EpsmBvhNode``: the boxes of up to four children in one 128-byte record, one
    component of all four per 16-byte quad, leaf children embedded).  Returns a dict:

      nodes        (n,32) float32; columns 0..23 lox loy loz hix hiy hiz (4 each), 24..27 the child references and
                   28..31 the leaf counts as int32 bits
      order        (T,)   triangle ids in leaf order (``prim_index``)
      leaf_node / leaf_slot / leaf_tris   which (node, child slot) is a leaf and its triangles as rows of
                   leaf-ordered indices, padded to ``leaf_size`` by repetition          -> refit step 1
      levels       [(node, slot, child), ...] inner child slots grouped by depth, deepest first -> refit step 2
    """
    T = tri.shape[0]
    p = pos[tri]                                   # (T,3,3)
    lo_t, hi_t = p.min(axis=1), p.max(axis=1)
    cen = 0.5 * (lo_t + hi_t)
    order = np.arange(T, dtype=np.int64)
    # binary tree first: node = [a, b, left, right] over order[a:b]; leaves have left = -1
    tree: List[list] = [[0, T, -1, -1]]
    tdepth = [0]
    stack = [0]
    while stack:
        ni = stack.pop()
        a, b = tree[ni][0], tree[ni][1]
        if b - a <= leaf_size:
            continue
        ids = order[a:b]
        c = cen[ids]
        mid = None
        # A SAH plane may peel off a few triangles per level (mixed triangle scales: a teapot in a stadium); halving cannot.
        # SAH is used while the subtree can still be finished by halving inside kMaxBinaryHeight levels, which is what the
        # four-wide collapse below can always fold into the traversal's 16 wide levels.
        if b - a > sah_min and tdepth[ni] + 1 + int(np.ceil(np.log2(b - a))) <= kMaxBinaryHeight:
            split = _sah_split(c, lo_t[ids], hi_t[ids])
            if split is not None:
                left_mask = split
                mid = int(left_mask.sum())
                order[a:b] = np.concatenate([ids[left_mask], ids[~left_mask]])
        if mid is None:                            # small node or no useful plane: median of the widest axis
            ax = int(np.argmax(c.max(axis=0) - c.min(axis=0)))
            mid = (b - a) // 2
            order[a:b] = ids[np.argpartition(c[:, ax], mid)]
        tree[ni][2], tree[ni][3] = len(tree), len(tree) + 1
        tree.append([a, a + mid, -1, -1]); tree.append([a + mid, b, -1, -1])
        tdepth += [tdepth[ni] + 1, tdepth[ni] + 1]
        stack += [tree[ni][2], tree[ni][3]]
    # four-wide nodes: a wide node takes a binary inner node and, while it has fewer than four children and one of them
    # is inner, replaces the inner child with the largest box by that child's two children -- unless a child's subtree is
    # too TALL for the wide levels left below this node: then the tallest child is opened instead.  (Opening the tallest
    # child twice takes two binary levels off every chain, so a subtree of binary height h fits into ceil(h / 2) wide
    # levels whatever its shape; a wide node at depth d may therefore keep children of height <= 2 (kMaxWideDepth - 1 - d).)
    t_lo = np.zeros((len(tree), 3)); t_hi = np.zeros((len(tree), 3))
    for bi in range(len(tree) - 1, -1, -1):          # children follow their parent in `tree`
        a_, b_, l_, r_ = tree[bi]
        if l_ < 0:
            ids = order[a_:b_]
            t_lo[bi], t_hi[bi] = lo_t[ids].min(axis=0), hi_t[ids].max(axis=0)
        else:
            t_lo[bi], t_hi[bi] = np.minimum(t_lo[l_], t_lo[r_]), np.maximum(t_hi[l_], t_hi[r_])

    t_height = np.zeros(len(tree), dtype=np.int64)   # leaves 0
    for bi in range(len(tree) - 1, -1, -1):
        if tree[bi][2] >= 0:
            t_height[bi] = 1 + max(t_height[tree[bi][2]], t_height[tree[bi][3]])

    def half_area(bi):
        e = t_hi[bi] - t_lo[bi]
        return e[0] * e[1] + e[1] * e[2] + e[2] * e[0]

    wide_children: Dict[int, list] = {}
    wide_order: List[int] = []
    todo = [(0, 0)] if tree[0][2] >= 0 else []
    while todo:
        bi, wd = todo.pop(0)
        allowed = 2 * (kMaxWideDepth - 1 - wd)       # tallest subtree a child of this wide node may root
        kids = [tree[bi][2], tree[bi][3]]
        while len(kids) < 4:
            inner_kids = [k for k in kids if tree[k][2] >= 0]
            if not inner_kids:
                break
            too_tall = [k for k in inner_kids if t_height[k] > allowed]
            k = max(too_tall, key=lambda q: t_height[q]) if too_tall else max(inner_kids, key=half_area)
            i = kids.index(k)
            kids[i:i + 1] = [tree[k][2], tree[k][3]]
        wide_children[bi] = kids
        wide_order.append(bi)
        todo += [(k, wd + 1) for k in kids if tree[k][2] >= 0]
    wide_of = {bi: wi for wi, bi in enumerate(wide_order)}
    n = max(1, len(wide_order))
    nodes = np.zeros((n, 32), dtype=np.float32)      # EpsmBvhNode: lox loy loz hix hiy hiz (4 each) | c[4] | n[4]
    inodes = nodes.view(np.int32)
    nodes[:, 0:12] = np.inf; nodes[:, 12:24] = -np.inf
    inodes[:, 24:28] = 0x7fffffff                  # absent child
    leaf_node, leaf_slot, leaf_tris = [], [], []
    depth = np.zeros(n, dtype=np.int64)
    per_level: Dict[int, list] = {}

    def put_child(wi, slot, bi):
        a, b, left, _ = tree[bi]
        if left < 0:
            inodes[wi, 24 + slot], inodes[wi, 28 + slot] = ~((a << 3) | (b - a)), b - a     # leaf reference
            leaf_node.append(wi); leaf_slot.append(slot)
            leaf_tris.append([a + min(j, b - a - 1) for j in range(leaf_size)])
        else:
            ci = wide_of[bi]
            inodes[wi, 24 + slot], inodes[wi, 28 + slot] = ci, 0
            depth[ci] = depth[wi] + 1
            per_level.setdefault(int(depth[wi]), []).append((wi, slot, ci))

    if not wide_order:
        put_child(0, 0, 0)
    for bi in wide_order:                          # breadth first: parents precede their children
        for slot, k in enumerate(wide_children[bi]):
            put_child(wide_of[bi], slot, k)
    if int(depth.max(initial=0)) + 1 > kMaxWideDepth:      # (cannot happen: both bounds above hold by construction)
        raise ValueError("BVH deeper than the traversal stack (kBvhStack = 3 pushes x 16 levels)")
    levels = [np.asarray(per_level[d], dtype=np.int64).reshape(-1, 3) for d in sorted(per_level, reverse=True)]
    return {"nodes": nodes, "order": order.astype(np.int64),
            "leaf_node": np.asarray(leaf_node, np.int64), "leaf_slot": np.asarray(leaf_slot, np.int64),
            "leaf_tris": np.asarray(leaf_tris, np.int64).reshape(-1, leaf_size), "levels": levels}


class DeviceBvh:
    """The BVH on the device + its refit: after the vertices moved, the boxes are recomputed bottom-up with a
    handful of torch gathers (leaf slots from the triangles, then one level of inner slots after the other);
    the topology of the build is kept (``params.update()`` per optimisation step must not cost a rebuild)."""

    def __init__(self, plan: dict, device):
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
        self.nodes = t(plan["nodes"]).clone()                      # (n,32) float32; refit() writes the boxes: never the (cached) plan's own array
        self.order = t(plan["order"])
        self.prim_index = self.order.to(torch.int32)
        self.leaf_node, self.leaf_slot, self.leaf_tris = t(plan["leaf_node"]), t(plan["leaf_slot"]), t(plan["leaf_tris"])
        self.levels = [(t(l[:, 0]), t(l[:, 1]), t(l[:, 2])) for l in plan["levels"]]
        self._cols = torch.arange(3, device=device) * 4            # component k of slot s sits in column 4 k + s (lo) / 12 + 4 k + s (hi)
        self.tri_verts = None

    def _write(self, node, slot, lo, hi):
        c = slot.unsqueeze(1) + self._cols
        self.nodes[node.unsqueeze(1), c] = lo
        self.nodes[node.unsqueeze(1), c + 12] = hi

    def refit(self, positions: torch.Tensor, tri: torch.Tensor):
        """positions (V,3) f32, tri (T,3) int -> tri_verts (T,9) in leaf order and fresh boxes."""
        tv = positions[tri.long()[self.order]]                     # (T,3,3)
        if self.tri_verts is None:
            self.tri_verts = tv.reshape(-1, 9).contiguous()
        else:
            self.tri_verts.copy_(tv.reshape(-1, 9))                # in place: the scene struct keeps its pointer
        if self.leaf_node.numel():
            g = tv[self.leaf_tris]                                 # (L, leaf, 3, 3)
            lo, hi = g.amin(dim=(1, 2)), g.amax(dim=(1, 2))
            self._write(self.leaf_node, self.leaf_slot, lo - 1e-6 * (1 + lo.abs()), hi + 1e-6 * (1 + hi.abs()))
        for node, slot, child in self.levels:
            cb = self.nodes[child]                                 # union of the child's (up to) four boxes; absent: +-inf
            self._write(node, slot, cb[:, 0:12].reshape(-1, 3, 4).amin(dim=2), cb[:, 12:24].reshape(-1, 3, 4).amax(dim=2))


# ---------------------------------------------------------------------------- scene
def _rgb(x, default):
    if x is None:
        return np.array(default, dtype=np.float32)
    if isinstance(x, dict):
        x = x.get("value", default)
    a = np.asarray(x, dtype=np.float32).reshape(-1)
    return np.repeat(a, 3) if a.size == 1 else a[:3]


def _bitmap_texture(val: dict, base_dir: str) -> dict:
    """A ``bitmap`` texture (src/textures/bitmap.cpp) as linear RGB: ``bitmap`` / ``data`` (an array) or ``filename`` (.npy --
    no image readers here; the reference's .jpg / .png textures are not part of the repository), ``filter_type`` bilinear
    (default) or nearest, ``wrap_mode`` repeat."""
    a = val.get("bitmap", val.get("data"))
    if a is None:
        fn = val.get("filename")
        if fn is None or not fn.endswith(".npy"):
            raise ValueError(f"bitmap: give 'bitmap' (an array) or a .npy 'filename' (got {fn!r}: no image readers here)")
        a = np.load(os.path.join(base_dir, fn))
    a = np.asarray(a.detach().cpu().numpy() if torch.is_tensor(a) else a, dtype=np.float32)
    if a.ndim == 2:
        a = np.repeat(a[:, :, None], 3, axis=2)
    if a.ndim != 3 or a.shape[2] < 3:
        raise ValueError("bitmap: the array must be (H, W) or (H, W, 3)")
    if val.get("wrap_mode", "repeat") != "repeat":
        raise ValueError("bitmap: only wrap_mode 'repeat'")
    if "to_uv" in val:
        raise ValueError("bitmap: to_uv is not supported (scale the mesh's texture coordinates instead)")
    ft = val.get("filter_type", "bilinear")
    if ft not in ("bilinear", "nearest"):
        raise ValueError("bitmap: filter_type 'bilinear' or 'nearest'")
    return {"bitmap": np.ascontiguousarray(a[:, :, :3]), "nearest": 1 if ft == "nearest" else 0}


def _envmap_bitmap(val: dict, base_dir: str) -> np.ndarray:
    """(H, W, 3) float32 radiance of an ``envmap`` emitter, ``scale`` applied: ``bitmap`` / ``data`` (an array, as
    ``mi.Bitmap(array)`` would carry it) or ``filename`` (.npy -- there is no OpenEXR reader here; the reference's experiments
    load .exr files that are not part of the repository, EPSM/exp/glossyball.py:107-108)."""
    a = val.get("bitmap", val.get("data"))
    if a is None:
        fn = val.get("filename")
        if fn is None:
            raise ValueError("envmap: give 'bitmap' (an array) or 'filename' (.npy)")
        if not fn.endswith(".npy"):
            raise ValueError(f"envmap: cannot read {fn!r}: only .npy arrays (H, W, 3) are supported (no OpenEXR reader)")
        a = np.load(os.path.join(base_dir, fn))
    a = np.asarray(a.detach().cpu().numpy() if torch.is_tensor(a) else a, dtype=np.float32)
    if a.ndim == 2:
        a = np.repeat(a[:, :, None], 3, axis=2)
    if a.ndim != 3 or a.shape[2] < 3 or a.shape[0] < 2 or a.shape[1] < 2:
        raise ValueError("envmap: the bitmap must be (H >= 2, W >= 2, 3)")
    return np.ascontiguousarray(a[:, :, :3] * float(val.get("scale", 1.0)))


def environment_tables(bitmap: np.ndarray):
    """The arrays of ``EpsmEnvironment`` for an (H, W, 3) lat-long map: texels (H, W + 1, 3), column W a copy of column 0 (the
    extra column envmap.cpp:205-229 appends, so that the interpolation wraps), and the piecewise-constant
    sampling distribution over the W x (H - 1) bilinear cells -- weight = mean over a cell's corners of luminance x sin(theta)
    (envmap.cpp:268-296 weighs texels the same way; its hierarchical warp follows the bilinear interpolant, this one is constant
    per cell).  Returns texels, row_cdf (H - 1), col_cdf (H - 1, W), cell_pdf (H - 1, W) in (u, v) in [0, 1)^2."""
    H, W = bitmap.shape[:2]
    t = np.concatenate([bitmap, bitmap[:, :1]], axis=1).astype(np.float64)
    lum = (0.212671 * t[..., 0] + 0.715160 * t[..., 1] + 0.072169 * t[..., 2]) * np.sin(np.arange(H) * math.pi / (H - 1))[:, None]
    w = 0.25 * (lum[:-1, :-1] + lum[:-1, 1:] + lum[1:, :-1] + lum[1:, 1:])             # (H - 1, W)
    total = float(w.sum())
    rows = w.sum(1)
    row_cdf = np.cumsum(rows) / total if total > 0 else np.linspace(1.0 / (H - 1), 1.0, H - 1)
    col_cdf = np.cumsum(w, axis=1) / np.where(rows > 0, rows, 1.0)[:, None]
    col_cdf[rows <= 0] = np.linspace(1.0 / W, 1.0, W)
    row_cdf[-1] = 1.0; col_cdf[:, -1] = 1.0
    cell_pdf = w / total * (W * (H - 1)) if total > 0 else np.zeros_like(w)
    f = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    return f(t), f(row_cdf), f(col_cdf), f(cell_pdf)


def _ior(x, default):
    if x is None:
        return default
    return IOR[x] if isinstance(x, str) else float(x)


class Sensor:
    def __init__(self, d: dict):
        film = next((v for v in d.values() if isinstance(v, dict) and v.get("type") == "hdrfilm"), {})
        sampler = next((v for v in d.values() if isinstance(v, dict) and v.get("type") == "independent"), {})
        # the film and its crop window (hdrfilm: crop_width / crop_height / crop_offset_x / crop_offset_y; src/render/film.cpp): the
        # image, the wavefront and the position differentials are those of the WINDOW (m_resolution = crop_size,
        # perspective.cpp:172-183); the full film only fixes the aspect ratio and where the window sits
        self.film_width, self.film_height = int(film.get("width", 768)), int(film.get("height", 576))
        self.width, self.height = int(film.get("crop_width", self.film_width)), int(film.get("crop_height", self.film_height))
        self.crop_offset = (int(film.get("crop_offset_x", 0)), int(film.get("crop_offset_y", 0)))
        if (self.width < 1 or self.height < 1 or min(self.crop_offset) < 0 or self.crop_offset[0] + self.width > self.film_width
                or self.crop_offset[1] + self.height > self.film_height):
            raise ValueError("film: invalid crop window")          # film.cpp: "Invalid crop window specification!"
        rf = film.get("rfilter", {"type": "gaussian"})
        self.rfilter = {"box": 0, "gaussian": 1}[rf.get("type", "gaussian")]
        # film.sample_border (hdrfilm): samples are also generated in a border of rfilter.border_size() pixels around the
        # film (ceil(radius - 1/2): 2 for the gaussian of radius 2, 0 for the box) so that what enters or leaves the
        # viewport is accounted for (common.py:309-336, 390-399); the sensors the reparameterised integrator renders with set it
        self.sample_border = bool(film.get("sample_border", False))
        self.border = 2 if (self.sample_border and self.rfilter == 1) else 0
        self.spp = int(sampler.get("sample_count", 4))
        self.fov = float(d.get("fov", 45.0))
        self.near, self.far = float(d.get("near_clip", 1e-2)), float(d.get("far_clip", 1e4))
        self.to_world = np.asarray(d.get("to_world", np.eye(4)), dtype=np.float64)

    def wavefront_size(self, spp: int) -> int:
        """Paths of one pass: every pixel of the film (and of its sample border) spp times."""
        return (self.width + 2 * self.border) * (self.height + 2 * self.border) * int(spp)

    def c_struct(self) -> EpsmSensor:
        key = (self.width, self.height, self.film_width, self.film_height, self.crop_offset, self.fov, self.near, self.far,
               self.to_world.tobytes(), self.border)
        if getattr(self, "_c_key", None) == key:
            return self._c_struct
        s = EpsmSensor()
        cropped = (self.width, self.height) != (self.film_width, self.film_height) or self.crop_offset != (0, 0)
        c2s = perspective_projection(self.film_width, self.film_height, self.fov, self.near, self.far,
                                     (self.width, self.height) + tuple(self.crop_offset) if cropped else None)
        s2c = np.linalg.inv(c2s)
        s.to_world[:] = self.to_world[:3, :].astype(np.float32).reshape(-1).tolist()
        s.sample_to_camera[:] = s2c.astype(np.float32).reshape(-1).tolist()
        p0 = _xform_point(s2c, [0, 0, 0])
        s.dx[:] = (_xform_point(s2c, [1.0 / self.width, 0, 0]) - p0).astype(np.float32).tolist()    # perspective.cpp:178-182
        s.dy[:] = (_xform_point(s2c, [0, 1.0 / self.height, 0]) - p0).astype(np.float32).tolist()
        s.near_clip, s.far_clip, s.width, s.height = self.near, self.far, self.width, self.height
        s.border, s.pad = self.border, 0
        self._c_key, self._c_struct = key, s
        return s


_BVH_CACHE: Dict[bytes, object] = {}        # geometry fingerprint -> host-side tree (build_bvh), a handful of entries


class Scene:
    """Flattened scene + device buffers.  Implements the tracer protocol the integrators use:
    ``trace_paths`` (records for the backward pass) and ``render_primal``."""

    def __init__(self, meshes: Sequence[Mesh], bsdfs: Sequence[dict], emitters: Sequence[dict],
                 sensors: Sequence[Sensor], device="cuda", bsdf_names: Optional[Sequence[str]] = None,
                 tile_paths: int = _dist.TILE_PATHS):
        self.meshes, self.bsdf_desc, self.emitter_desc, self.sensors = list(meshes), list(bsdfs), list(emitters), list(sensors)
        self.bsdf_names = list(bsdf_names or [f"bsdf{i}" for i in range(len(bsdfs))])
        self.device = torch.device(device)
        self._tile_paths = int(tile_paths)
        self.tile_paths_explicit = False           # set when the caller assigns Scene.tile_paths: then it is an UPPER bound everywhere
        self.alpha_slots: Dict[int, int] = {}
        self.color_slots: List[tuple] = []         # colour parameters attached for the colour adjoint: ("bsdf" | "emitter", index)
        self.rr_depth = 5
        # test hook: tests/host_harness compiles the tracer's per-path code for the CPU and plugs
        # its entry points in here; the product path (None) is the HIP library and needs a GPU
        self._backend = None
        # "mega": one launch, a lane carries its path through all bounces; "wavefront": queues of live paths,
        # three kernels per bounce; "auto": wavefront from WAVEFRONT_MIN_TRIANGLES triangles on (where the
        # one-launch form is bound by divergence; below, the wavefront's state traffic costs more than it saves)
        self.tracer = "auto"
        # where the native log's rays and records lie (records.alloc_log): "dense" = two arrays; "interleaved" = one block of
        # K + 1 cache lines per path (ABI v7: 16 % fewer bytes read by the backward kernel, but slower where it counts --
        # MEASUREMENTS.md 10.12)
        self.log_layout = "dense"
        self._wf_workspace = {}        # scratch of the wavefront tracer, one per stream it was used on
        self._upload()

    @property
    def tile_paths(self) -> int:
        """Paths per tile of a pass (the sharding unit).  The wavefront tracer works in tiles of up to WAVEFRONT_TILE_PATHS
        whatever this says (its workspace is 236 B per path); prb_reparam raises the DEFAULT to as much as the sharding allows
        (7 GB of warp requests at 2^23 paths) but never exceeds a value the caller ASSIGNED -- their memory bound."""
        return self._tile_paths

    @tile_paths.setter
    def tile_paths(self, n: int) -> None:
        self._tile_paths = int(n)
        self.tile_paths_explicit = True

    # -- construction from the reference's dict shape ---------------------------------------
    @staticmethod
    def from_dict(d: dict, device="cuda", base_dir: str = ".") -> "Scene":
        assert d.get("type") == "scene"
        bsdfs, bsdf_names, named = [], [], {}

        def parse_bsdf(b: dict) -> dict:
            t = b["type"]
            if t == "twosided":
                inner = next(v for v in b.values() if isinstance(v, dict) and "type" in v)
                out = parse_bsdf(inner); out["twosided"] = 1
                return out
            o = dict(type=BSDF_TYPES[t], twosided=0, distr=0, sample_visible=1, reflectance=_rgb(None, [1, 1, 1]),
                     alpha=0.1, eta=np.zeros(3, np.float32), k=np.ones(3, np.float32), int_ior=1.5046, ext_ior=1.000277)
            if t == "diffuse":
                r = b.get("reflectance")
                if isinstance(r, dict) and r.get("type") == "bitmap":
                    o["texture"] = _bitmap_texture(r, base_dir)
                    o["reflectance"] = o["texture"]["bitmap"].reshape(-1, 3).mean(0).astype(np.float32)      # (a stand-in: the tracer looks the texture up)
                else:
                    o["reflectance"] = _rgb(r, [0.5, 0.5, 0.5])
            elif t in ("conductor", "roughconductor"):
                mat = b.get("material", "Cu")
                if "eta" in b or "k" in b:
                    o["eta"], o["k"] = _rgb(b.get("eta"), [0, 0, 0]), _rgb(b.get("k"), [1, 1, 1])
                else:
                    e, k = CONDUCTORS[mat]
                    o["eta"], o["k"] = np.array(e, np.float32), np.array(k, np.float32)
                o["reflectance"] = _rgb(b.get("specular_reflectance"), [1, 1, 1])
                if t == "roughconductor":
                    o["distr"] = {"beckmann": 0, "ggx": 1}[b.get("distribution", "beckmann")]
                    a = b.get("alpha", 0.1)
                    o["alpha"] = float(a["value"] if isinstance(a, dict) else a)
                    o["sample_visible"] = 1 if b.get("sample_visible", True) else 0
            elif t == "dielectric":
                o["int_ior"], o["ext_ior"] = _ior(b.get("int_ior"), 1.5046), _ior(b.get("ext_ior"), 1.000277)
                o["reflectance"] = _rgb(b.get("specular_reflectance"), [1, 1, 1])
            return o

        def add_bsdf(name, b):
            bsdfs.append(parse_bsdf(b)); bsdf_names.append(name)
            return len(bsdfs) - 1

        for key, val in d.items():
            if isinstance(val, dict) and val.get("type") in list(BSDF_TYPES) + ["twosided"]:
                named[val.get("id", key)] = add_bsdf(key, val)
                named[key] = named[val.get("id", key)]
        default_bsdf = None
        meshes, emitters, sensors = [], [], []
        for key, val in d.items():
            if not isinstance(val, dict):
                continue
            t = val.get("type")
            if t == "perspective":
                sensors.append(Sensor(val))
            elif t == "point":
                emitters.append(dict(type=1, mesh=-1, radiance=_rgb(val.get("intensity"), [1, 1, 1]),
                                     position=np.asarray(val.get("position", [0, 0, 0]), np.float32)))
            elif t in ("constant", "envmap"):
                if any(e["type"] in (2, 3) for e in emitters):
                    raise ValueError("a scene has at most one environment emitter")
                e = dict(type=2 if t == "constant" else 3, mesh=-1, radiance=_rgb(val.get("radiance"), [1, 1, 1]),
                         position=np.zeros(3, np.float32))
                if t == "envmap":
                    e["bitmap"] = _envmap_bitmap(val, base_dir)
                    e["to_world"] = np.asarray(val.get("to_world", np.eye(4)), dtype=np.float64)
                emitters.append(e)
            elif t in ("obj", "ply", "mesh", "rectangle"):
                tw = np.asarray(val.get("to_world", np.eye(4)), dtype=np.float64)
                uv = None
                if t == "obj":
                    v, n, f, uv = load_obj(os.path.join(base_dir, val["filename"]), with_uv=True)
                    if uv is None:                                       # (no vt lines: the plain reader's vertex merging)
                        v, n, f = load_obj(os.path.join(base_dir, val["filename"]))
                elif t == "ply":
                    v, n, f = load_ply(os.path.join(base_dir, val["filename"]))
                elif t == "mesh":
                    v, f = np.asarray(val["vertices"], np.float64), np.asarray(val["faces"], np.int64)
                    n = np.asarray(val["normals"], np.float64) if "normals" in val else None
                    uv = np.asarray(val["texcoords"], np.float64) if "texcoords" in val else None
                else:
                    v = np.array([[-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0]], np.float64)
                    f = np.array([[0, 1, 2], [0, 2, 3]], np.int64); n = None
                    uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float64)                # rectangle.cpp: uv over [0,1]^2
                v = (tw[:3, :3] @ v.T).T + tw[:3, 3]
                if n is not None:
                    nt = np.linalg.inv(tw[:3, :3]).T
                    n = (nt @ n.T).T; n /= np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-30)
                bsdf_id, emitter_id = None, -1
                for k2, v2 in val.items():
                    if not isinstance(v2, dict):
                        continue
                    if v2.get("type") == "ref":
                        bsdf_id = named[v2["id"]]
                    elif v2.get("type") in list(BSDF_TYPES) + ["twosided"]:
                        bsdf_id = add_bsdf(f"{key}.{k2}", v2)
                    elif v2.get("type") == "area":
                        emitters.append(dict(type=0, mesh=len(meshes), radiance=_rgb(v2.get("radiance"), [1, 1, 1]),
                                             position=np.zeros(3, np.float32)))
                        emitter_id = len(emitters) - 1
                if bsdf_id is None:
                    if default_bsdf is None:
                        default_bsdf = add_bsdf("__default_diffuse", {"type": "diffuse"})
                    bsdf_id = default_bsdf
                face_normals = bool(val.get("face_normals", False)) or t == "rectangle"
                meshes.append(Mesh(key, v, f, n, bsdf=bsdf_id, emitter=emitter_id,
                                   flip_normals=bool(val.get("flip_normals", False)), is_mesh=(t != "rectangle"),
                                   face_normals=face_normals, uv=uv))
        return Scene(meshes, bsdfs, emitters, sensors, device=device, bsdf_names=bsdf_names)

    # -- parameters ----------------------------------------------------------------------------
    def mesh(self, name: str) -> Mesh:
        return next(m for m in self.meshes if m.name == name)

    def attach(self, mesh_name: str, positions: bool = True, normals: bool = False):
        """``dr.enable_grad(params['<mesh>.vertex_positions' / '.vertex_normals'])``."""
        m = self.mesh(mesh_name)
        m.pos_attached, m.nrm_attached = positions, normals
        self._refresh_attach_flags()

    def _refresh_attach_flags(self, sync_host: bool = True):
        """What `dr.enable_grad` changes on the device: the mode bits of the meshes -- in the mesh table the tracer reads and
        in the last column of the triangle table the gradient kernels read.  Rewritten IN PLACE (the scene struct's pointers
        stay valid); nothing else of the upload is repeated."""
        if (getattr(self, "_mesh_structs", None) is None or getattr(self, "tri_table", None) is None
                or (sync_host and any(getattr(m, "host_stale", False) for m in self.meshes))):   # (vertices moved on the device: the full upload syncs the host copies)
            return self._upload()
        for c, m in zip(self._mesh_structs, self.meshes):
            c.flags = m.flags()
        self._mesh_buf.copy_(torch.frombuffer(bytearray(bytes(self._mesh_structs)), dtype=torch.uint8))
        if self.T > 0:
            mode = torch.tensor([(m.flags() & 0xF) | ((self.alpha_slots.get(m.bsdf, -1) + 1) << 8) for m in self.meshes],
                                dtype=torch.int32, device=self.device)
            self.tri_table[:, 3] = mode[self.tri_mesh.long()]

    def attach_sensor(self, attached: bool = True):
        """``dr.enable_grad(params['sensor.to_world'])`` for its translation: ``prb_reparam``'s render_backward then leaves
        d loss / d (world-space position of the sensor) in ``ParamGrads.cam_origin`` (test_ad_integrators.py:639-674,
        TranslateCameraConfig; EPSM/exp/bedroom.py:18-36 optimises the camera in the hybrid scheme)."""
        self.sensor_attached = bool(attached)

    def has_attached_geometry(self) -> bool:
        """Any mesh whose vertex positions / normals receive gradients?"""
        return any(getattr(m, "pos_attached", False) or getattr(m, "nrm_attached", False) for m in self.meshes)

    def attach_alpha(self, bsdf_name: str) -> int:
        i = self.bsdf_names.index(bsdf_name)
        self.alpha_slots.setdefault(i, len(self.alpha_slots))
        self._upload()
        return self.alpha_slots[i]

    def attach_color(self, bsdf_name: str) -> int:
        """``dr.enable_grad(params['<bsdf>.reflectance.value'])`` for the colour adjoint (PRBIntegrator): diffuse BSDFs."""
        i = self.bsdf_names.index(bsdf_name)
        if self.bsdf_desc[i]["type"] != 0 or "texture" in self.bsdf_desc[i]:
            raise ValueError("attach_color: only the CONSTANT reflectance of a diffuse BSDF is a colour parameter here")
        if ("bsdf", i) not in self.color_slots:
            if len(self.color_slots) >= 4:
                raise ValueError("at most 4 colour parameters")
            self.color_slots.append(("bsdf", i))
            self._upload()
        return self.color_slots.index(("bsdf", i))

    def attach_radiance(self, mesh_or_index) -> int:
        """``dr.enable_grad(params['<emitter>.radiance.value'])``: by emitting mesh name (area light) or emitter index."""
        i = mesh_or_index if isinstance(mesh_or_index, int) else self.mesh(mesh_or_index).emitter
        if not (0 <= i < len(self.emitter_desc)):
            raise ValueError("attach_radiance: not an emitter")
        if self.emitter_desc[i]["type"] == 3:
            raise ValueError("attach_radiance: an envmap's radiance is its bitmap (not a colour parameter here); a `constant` emitter's is")
        if ("emitter", i) not in self.color_slots:
            if len(self.color_slots) >= 4:
                raise ValueError("at most 4 colour parameters")
            self.color_slots.append(("emitter", i))
            self._upload()
        return self.color_slots.index(("emitter", i))

    def color_values(self) -> torch.Tensor:
        """(C,3) current values of the attached colour parameters, slot order."""
        rows = [self.bsdf_desc[i]["reflectance"] if kind == "bsdf" else self.emitter_desc[i]["radiance"] for kind, i in self.color_slots]
        return torch.tensor([[float(x) for x in r] for r in rows], dtype=torch.float32, device=self.device).reshape(-1, 3)

    def set_color(self, slot: int, rgb):
        """``params[...] = rgb; params.update()`` for colour slot ``slot``."""
        kind, i = self.color_slots[slot]
        (self.bsdf_desc[i] if kind == "bsdf" else self.emitter_desc[i])["reflectance" if kind == "bsdf" else "radiance"] = [float(x) for x in rgb]
        self._upload()

    def set_alpha(self, bsdf_name: str, alpha: float):
        """``params['<bsdf>.alpha.value'] = alpha; params.update()``: only the BSDF table is rewritten (in place)."""
        self.bsdf_desc[self.bsdf_names.index(bsdf_name)]["alpha"] = float(alpha)
        self._bsdf_buf.copy_(torch.frombuffer(bytearray(bytes(self._bsdf_structs())), dtype=torch.uint8))

    def _bsdf_structs(self):
        bs = (EpsmBsdf * max(1, len(self.bsdf_desc)))()
        for i, b in enumerate(self.bsdf_desc):
            c = bs[i]
            c.type, c.twosided, c.distr, c.sample_visible = b["type"], b["twosided"], b["distr"], b["sample_visible"]
            c.reflectance[:] = [float(x) for x in b["reflectance"]]
            c.alpha = float(b["alpha"])
            c.eta[:] = [float(x) for x in b["eta"]]; c.k[:] = [float(x) for x in b["k"]]
            c.int_ior, c.ext_ior = float(b["int_ior"]), float(b["ext_ior"])
            c.alpha_slot = self.alpha_slots.get(i, -1)
            c.color_slot = self.color_slots.index(("bsdf", i)) if ("bsdf", i) in self.color_slots else -1
            c.texture = b.get("texture_index", -1)
        return bs

    def set_vertex_positions(self, mesh_name: str, v):
        """``params['<mesh>.vertex_positions'] = v; params.update()``: everything stays on the device -- the
        rows of the flat position buffer are overwritten, vertex normals are recomputed like
        Mesh::parameters_changed does, the BVH is refitted (same topology, fresh boxes).  Only an emitting
        mesh (its area / sampling CDF change) goes through the full host-side upload."""
        m = self.mesh(mesh_name)
        lo, hi = self.mesh_slices[mesh_name]
        if m.emitter >= 0 or self.bvh is None:
            m.v = np.asarray(v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v, dtype=np.float64).reshape(-1, 3)
            if m.has_normals:
                m.n = vertex_normals(m.v, m.f)
            self._upload()
            return
        v_t = torch.as_tensor(v, dtype=torch.float32, device=self.device).detach().reshape(-1, 3)
        if v_t.shape[0] != hi - lo:
            raise ValueError(f"{mesh_name}: expected {hi - lo} vertices, got {v_t.shape[0]}")
        self.positions[lo:hi] = v_t
        m.host_stale = True                                   # m.v / m.n are refreshed when the host needs them
        if m.has_normals:
            t0, t1 = self.mesh_tri_slices[mesh_name]
            self.normals[lo:hi] = vertex_normals_torch(v_t, self.tri[t0:t1].long() - lo)
        self.bvh.refit(self.positions, self.tri)

    def _sync_host_meshes(self):
        for m in self.meshes:
            if getattr(m, "host_stale", False):
                lo, hi = self.mesh_slices[m.name]
                m.v = self.positions[lo:hi].detach().cpu().double().numpy()
                if m.has_normals:
                    m.n = self.normals[lo:hi].detach().cpu().double().numpy()
                m.host_stale = False

    def vertex_positions(self, mesh_name: str) -> torch.Tensor:
        lo, hi = self.mesh_slices[mesh_name]
        return self.positions[lo:hi]

    def vertex_normals(self, mesh_name: str) -> torch.Tensor:
        lo, hi = self.mesh_slices[mesh_name]
        return self.normals[lo:hi]

    def param_grads(self) -> ParamGrads:
        return ParamGrads(self.V, len(self.alpha_slots), device=self.device, mesh_slices=self.mesh_slices,
                          n_colors=len(self.color_slots))

    # -- upload ----------------------------------------------------------------------------------
    def _upload(self):
        dev = self.device
        if getattr(self, "positions", None) is not None:
            self._sync_host_meshes()
        pos, nrm, tri, tri_mesh, cdf = [], [], [], [], []
        self.mesh_slices, self.mesh_tri_slices = {}, {}
        mesh_c = (EpsmMesh * max(1, len(self.meshes)))()
        voff = toff = coff = 0
        for mi_, m in enumerate(self.meshes):
            nv, nt = m.v.shape[0], m.f.shape[0]
            self.mesh_slices[m.name] = (voff, voff + nv)
            self.mesh_tri_slices[m.name] = (toff, toff + nt)
            pos.append(m.v); nrm.append(m.n if m.n is not None else np.zeros_like(m.v))
            tri.append(m.f + voff); tri_mesh.append(np.full(nt, mi_, np.uint32))
            p = m.v[m.f]
            areas = 0.5 * np.linalg.norm(np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0]), axis=1)
            c = mesh_c[mi_]
            c.tri_begin, c.tri_count, c.flags, c.bsdf, c.emitter = toff, nt, m.flags(), m.bsdf, m.emitter
            c.area, c.cdf_begin = float(areas.sum()), coff
            cdf.append(np.cumsum(areas) / max(areas.sum(), 1e-30))
            voff += nv; toff += nt; coff += nt
        self.V, self.T = voff, toff
        P = np.concatenate(pos) if pos else np.zeros((0, 3))
        TRI = np.concatenate(tri) if tri else np.zeros((0, 3), np.int64)
        f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
        self.positions = f32(P)
        self.normals = f32(np.concatenate(nrm) if nrm else np.zeros((0, 3)))
        self.tri = torch.from_numpy(np.ascontiguousarray(TRI, dtype=np.int32)).to(dev)
        self.tri_mesh = torch.from_numpy(np.ascontiguousarray(np.concatenate(tri_mesh) if tri_mesh else np.zeros(0), dtype=np.int32)).to(dev)
        self.emitter_cdf = f32(np.concatenate(cdf) if cdf else np.zeros(1))
        # the triangle table of include/epsm.h: row t = [v0, v1, v2, EPSM_MODE_* of the owning mesh]; the tracer logs
        # triangle ids, the gradient kernels look the vertex rows up here
        # (+ bits 8..: alpha slot of the mesh's BSDF + 1, which the packed log does not carry per vertex)
        mode = np.array([(m.flags() & 0xF) | ((self.alpha_slots.get(m.bsdf, -1) + 1) << 8) for m in self.meshes], dtype=np.int64)
        tm = np.concatenate(tri_mesh).astype(np.int64) if tri_mesh else np.zeros(0, np.int64)
        table = np.concatenate([TRI.reshape(-1, 3), mode[tm].reshape(-1, 1) if len(tm) else np.zeros((0, 1), np.int64)], axis=1)
        if table.shape[0] == 0:
            table = np.zeros((1, 4), np.int64)
        self.tri_table = torch.from_numpy(np.ascontiguousarray(table, dtype=np.int32)).to(dev)
        self.bvh = None
        if self.T > 0:
            # the tree's topology depends on the geometry alone: attach() / attach_alpha() / set_color() come back here with the
            # same triangles and must not pay the host-side build again (9 s at 128 k triangles; round 4's bench spent 100 s
            # attaching 34 meshes), nor does a second Scene over the same geometry (bench.py's legs)
            key = hashlib.blake2b(P.tobytes() + TRI.tobytes(), digest_size=16).digest()
            host = _BVH_CACHE.get(key)
            if host is None:
                host = build_bvh(P, TRI)
                if len(_BVH_CACHE) >= 4:
                    _BVH_CACHE.pop(next(iter(_BVH_CACHE)))
                _BVH_CACHE[key] = host
            self.bvh = DeviceBvh(host, dev)
            self.bvh.refit(self.positions, self.tri)
        # texture coordinates (rows of the meshes that have none stay zero: they are not flagged EPSM_MESH_HAS_UV) and textures
        self.texcoords = None
        if any(m.uv is not None for m in self.meshes):
            uvs = [m.uv if m.uv is not None else np.zeros((m.v.shape[0], 2)) for m in self.meshes]
            for m, u in zip(self.meshes, uvs):
                if u.shape[0] != m.v.shape[0]:
                    raise ValueError(f"mesh {m.name}: {u.shape[0]} texture coordinates for {m.v.shape[0]} vertices")
            self.texcoords = f32(np.concatenate(uvs))
        self._tex_buf = []
        for b in self.bsdf_desc:
            if "texture" in b:
                b["texture_index"] = len(self._tex_buf)
                self._tex_buf.append((f32(b["texture"]["bitmap"]), b["texture"]["nearest"]))
        tx = (EpsmTexture * max(1, len(self._tex_buf)))()
        for i, (t_, nearest) in enumerate(self._tex_buf):
            tx[i].texels, tx[i].height, tx[i].width, tx[i].nearest = t_.data_ptr(), int(t_.shape[0]), int(t_.shape[1]), int(nearest)
        bs = self._bsdf_structs()
        em = (EpsmEmitter * max(1, len(self.emitter_desc)))()
        for i, e in enumerate(self.emitter_desc):
            c = em[i]
            c.type, c.mesh = e["type"], e["mesh"]
            c.radiance[:] = [float(x) for x in e["radiance"]]; c.position[:] = [float(x) for x in e["position"]]
            c.color_slot = self.color_slots.index(("emitter", i)) if ("emitter", i) in self.color_slots else -1
        as_dev = lambda arr: torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        self._mesh_buf, self._bsdf_buf, self._em_buf = as_dev(mesh_c), as_dev(bs), as_dev(em)
        self._mesh_structs = mesh_c                              # host copy: _refresh_attach_flags rewrites the flags in place
        self._tex_struct_buf = as_dev(tx)
        s = EpsmSceneC()
        s.positions, s.normals = self.positions.data_ptr(), self.normals.data_ptr()
        s.tri, s.tri_mesh = self.tri.data_ptr(), self.tri_mesh.data_ptr()
        s.meshes, s.n_meshes = self._mesh_buf.data_ptr(), len(self.meshes)
        s.bsdfs, s.n_bsdfs = self._bsdf_buf.data_ptr(), len(self.bsdf_desc)
        s.emitters, s.n_emitters = self._em_buf.data_ptr(), len(self.emitter_desc)
        s.emitter_cdf = self.emitter_cdf.data_ptr()
        if self.bvh is not None:
            s.bvh, s.n_nodes = self.bvh.nodes.data_ptr(), int(self.bvh.nodes.shape[0])
            s.prim_index, s.tri_verts = self.bvh.prim_index.data_ptr(), self.bvh.tri_verts.data_ptr()
        s.n_vertices, s.n_triangles = self.V, self.T
        s.texcoords = self.texcoords.data_ptr() if self.texcoords is not None else None
        s.textures, s.n_textures = self._tex_struct_buf.data_ptr(), len(self._tex_buf)
        # ---- the environment emitter: sampling tables + the bounding sphere of shapes and sensors (scene.cpp expands the
        #      scene's box by its sensors; constant.cpp:76-79, envmap.cpp:311-314)
        s.env.kind, s.env.emitter = 0, 0                        # EPSM_ENV_NONE
        for i, e in enumerate(self.emitter_desc):
            if e["type"] not in (2, 3):
                continue
            s.env.kind, s.env.emitter = e["type"] - 1, i       # EPSM_ENV_CONSTANT = 1, EPSM_ENV_ENVMAP = 2
            pts = [P.reshape(-1, 3)] if self.V else []
            pts.append(np.array([sn.to_world[:3, 3] for sn in self.sensors], np.float64).reshape(-1, 3))
            pts = np.concatenate(pts) if pts else np.zeros((1, 3))
            lo, hi = pts.min(0), pts.max(0)
            ctr = 0.5 * (lo + hi)
            s.env.center[:] = [float(x) for x in ctr]
            s.env.radius = float(max(8.9e-5, np.linalg.norm(hi - ctr) * (1.0 + 8.9e-5)))
            R = np.eye(3)
            if e["type"] == 3:
                R = np.asarray(e["to_world"], np.float64)[:3, :3]
                R = R / np.cbrt(abs(np.linalg.det(R)))
                if not np.allclose(R @ R.T, np.eye(3), atol=1e-5):
                    raise ValueError("envmap: to_world must be a rotation")
                tex, row_cdf, col_cdf, cell_pdf = environment_tables(e["bitmap"])
                self._env_buf = [torch.from_numpy(a).to(dev) for a in (tex, row_cdf, col_cdf, cell_pdf)]
                s.env.height, s.env.width = int(e["bitmap"].shape[0]), int(e["bitmap"].shape[1])
                s.env.texels, s.env.row_cdf, s.env.col_cdf, s.env.cell_pdf = (t.data_ptr() for t in self._env_buf)
            s.env.to_local[:] = [float(x) for x in R.T.reshape(-1)]       # world -> emitter frame
        self.c_scene = s

    # -- tracing ---------------------------------------------------------------------------------
    WAVEFRONT_MIN_TRIANGLES = 60000      # measured break-even on MI355X with the sparse log (64 k triangles: 4.02 / 3.93 ms per 4.2 M paths)
    # The wavefront form is 5 launches per bounce + 1 per tile and its later bounces carry a fraction of the paths (20 % at the fourth
    # bounce of the 128 k-triangle scene): tiles as large as the sharding over the ranks allows.  512x512 @ 64 spp on that
    # scene, whole gradient image: 26.7 / 21.6 / 18.8 / 16.9 / 16.7 ms with tiles of 2^20 .. 2^24 paths (0.9 KB of log and
    # workspace per path: 14.7 GB at 2^24, of 288).
    WAVEFRONT_TILE_PATHS = 1 << 24

    # ... and from WAVEFRONT_MIN_PATHS paths per launch on: below, a stage is as long as its slowest traversal whatever the queue
    # holds, and the one launch of the other form wins (128 k triangles, gradient-only trace: 0.45 / 0.68 ms at 2^19 paths, the
    # reference's own backward size; 0.62 / 0.80 at 2^20; 1.22 / 1.02 at 2^21 -- MEASUREMENTS.md 10.10)
    WAVEFRONT_MIN_PATHS = (1 << 20) + 1

    def use_wavefront(self, n_paths=None) -> bool:
        """The tracer form of a launch over ``n_paths`` paths (None: of a large one -- what sizes the tiles)."""
        if self.tracer not in ("auto", "mega", "wavefront"):
            raise ValueError("Scene.tracer must be 'auto', 'mega' or 'wavefront'")
        if self.tracer == "auto" and n_paths is not None and n_paths < self.WAVEFRONT_MIN_PATHS:
            return False
        return self.tracer == "wavefront" or (self.tracer == "auto" and self.T >= self.WAVEFRONT_MIN_TRIANGLES)

    def trace_color(self, sensor_index: int, seed: int, spp: int, max_depth: int, lo: int, hi: int):
        """``epsm_trace_paths_color``: film positions, radiance and the per-path colour sums (n, C, 3) of paths [lo, hi)."""
        dev = self.device
        if dev.type != "cuda" and self._backend is None:
            raise _lib.EpsmError("the tracer runs on the GPU only (no CPU fallback)")
        lib = self._backend if self._backend is not None else _lib.lib()
        stream = torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else None
        n, Cn = hi - lo, len(self.color_slots)
        if Cn == 0:
            raise ValueError("no colour parameter attached (Scene.attach_color / attach_radiance)")
        film_pos = torch.empty((n, 2), device=dev, dtype=torch.float32)
        radiance = torch.empty((n, 3), device=dev, dtype=torch.float32)
        valid = torch.empty((n,), device=dev, dtype=torch.uint8)
        sums = torch.empty((n, Cn, 3), device=dev, dtype=torch.float32)
        cs = self.sensors[sensor_index].c_struct()
        fn = lib.epsm_trace_paths_color
        fn.restype = C.c_int
        rc = fn(C.byref(self.c_scene), C.byref(cs), C.c_uint32(seed & 0xFFFFFFFF), int(spp), int(max_depth), int(self.rr_depth),
                C.c_int64(lo), C.c_int64(n), C.c_void_p(film_pos.data_ptr()), C.c_void_p(radiance.data_ptr()),
                C.c_void_p(valid.data_ptr()), C.c_void_p(sums.data_ptr()), int(Cn), C.c_void_p(stream))
        if rc != 0:
            _lib.check(rc, "epsm_trace_paths_color") if self._backend is None else (_ for _ in ()).throw(RuntimeError(f"host tracer rc={rc}"))
        return film_pos, radiance, sums

    def trace_reparam(self, sensor_index: int, seed: int, spp: int, max_depth: int, lo: int, hi: int, radiance, adj_radiance,
                      adj_film, grad_pos, grad_nrm, reparam_max_depth: int, reparam_rays: int, kappa: float, exponent: float,
                      antithetic: bool = False):
        """``epsm_trace_paths_reparam``: the reparameterised backward pass of paths [lo, hi) -- accumulates d loss / d vertex
        positions (and normals) into ``grad_pos`` / ``grad_nrm`` (V,3) given, per path, the radiance of the primal pass under
        the same seed, its adjoint and the adjoint of the film position + determinant (integrators.film_adjoint_reparam).
        ``antithetic``: EPSM_REPARAM_ANTITHETIC (auxiliary rays in mirrored pairs, reparam.py:82-84)."""
        dev = self.device
        if dev.type != "cuda" and self._backend is None:
            raise _lib.EpsmError("the tracer runs on the GPU only (no CPU fallback)")
        lib = self._backend if self._backend is not None else _lib.lib()
        stream = torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else None
        n = hi - lo
        for t_, w in ((radiance, 3), (adj_radiance, 3), (adj_film, 3)):
            assert t_.is_contiguous() and t_.dtype == torch.float32 and tuple(t_.shape) == (n, w) and t_.device.type == dev.type
        for t_ in (grad_pos, grad_nrm):
            assert t_.is_contiguous() and t_.dtype == torch.float32 and tuple(t_.shape) == (self.V, 3) and t_.device.type == dev.type
        cs = self.sensors[sensor_index].c_struct()
        wsb = lib.epsm_trace_reparam_workspace_bytes
        wsb.restype, wsb.argtypes = C.c_size_t, [C.c_int64]
        need = int(wsb(C.c_int64(n)))
        ws = getattr(self, "_reparam_ws", None)
        if ws is None or ws.numel() < max(need, 16) or ws.device != radiance.device:
            ws = self._reparam_ws = torch.empty(max(need, 16), device=radiance.device, dtype=torch.uint8)
        fn = lib.epsm_trace_paths_reparam
        fn.restype = C.c_int
        rc = fn(C.byref(self.c_scene), C.byref(cs), C.c_uint32(seed & 0xFFFFFFFF), int(spp), int(max_depth), int(self.rr_depth),
                C.c_int64(lo), C.c_int64(n), C.c_void_p(radiance.data_ptr()), C.c_void_p(adj_radiance.data_ptr()),
                C.c_void_p(adj_film.data_ptr()), int(reparam_max_depth), int(reparam_rays), C.c_float(kappa), C.c_float(exponent),
                C.c_uint32(1 if antithetic else 0), C.c_void_p(grad_pos.data_ptr()), C.c_void_p(grad_nrm.data_ptr()), C.c_void_p(ws.data_ptr()), C.c_size_t(ws.numel()),
                C.c_void_p(stream))
        if rc != 0:
            _lib.check(rc, "epsm_trace_paths_reparam") if self._backend is None else (_ for _ in ()).throw(RuntimeError(f"host tracer rc={rc}"))

    @staticmethod
    def _gradient_only_flags(gradient_only) -> int:
        """EPSM_TRACE_GRADIENT_ONLY (+ _CAUSTIC) for the variant named (None / False: the full trace)."""
        if not gradient_only:
            return 0
        if gradient_only not in ("manifold", "manifold_caustic"):
            raise ValueError(f"gradient_only: 'manifold', 'manifold_caustic' or None, got {gradient_only!r}")
        return EPSM_TRACE_GRADIENT_ONLY | (EPSM_TRACE_GRADIENT_CAUSTIC if gradient_only == "manifold_caustic" else 0)

    # The wavefront tracer hands the last few paths to ONE launch for the rest of their loop (include/epsm_trace.h,
    # EPSM_TRACE_NO_TAIL); False keeps the stages for every bounce (reports of the per-bounce queue lengths).
    wavefront_tail = True

    def _tail_flag(self) -> int:
        return 0 if self.wavefront_tail else EPSM_TRACE_NO_TAIL

    def wavefront_queue_lengths(self):
        """Paths alive into bounce 0..5 and visibility rays of bounce 0..5 of the LAST wavefront trace on the current stream
        (the counters the stages keep on the device; a host read, for reports).  Complete only with ``wavefront_tail = False``:
        the counters of the bounces the tail launch carried stay 0."""
        stream = torch.cuda.current_stream(self.device).cuda_stream if self.device.type == "cuda" else None
        ws = self._wf_workspace.get(stream)
        if ws is None:
            return None
        c = ws[:64].view(torch.int32).cpu().tolist()
        return {"alive": c[0:6], "shadow": c[8:14]}

    def _trace_packed(self, sensor_index: int, seed: int, spp: int, max_depth: int, K: int, lo: int, hi: int, gradient_only=None,
                      first_hit=None):
        """The same trace with the vertex log in the NATIVE layout of the backward kernel (EPSM_TRACE_PACKED_LOG,
        include/epsm.h EpsmPackedLog): the PathTrace carries ``log`` (a PackedLog) instead of per-field arrays."""
        from .integrators import PathTrace
        from .records import PackedLog, alloc_log
        dev = self.device
        if dev.type != "cuda" and self._backend is None:
            raise _lib.EpsmError("the tracer runs on the GPU only (no CPU fallback)")
        lib = self._backend if self._backend is not None else _lib.lib()
        stream = torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else None
        sensor = self.sensors[sensor_index]
        n = hi - lo
        assert K >= 1
        rays, verts = alloc_log(n, K, dev, self.log_layout)
        # a gradient-only trace forms no image: radiance / film positions / valid are not asked for, and nothing is written for them
        want_image = not gradient_only
        radiance = torch.empty((n, 3), device=dev, dtype=torch.float32) if want_image else None
        film_pos = torch.empty((n, 2), device=dev, dtype=torch.float32) if want_image else None
        valid = torch.empty((n,), device=dev, dtype=torch.uint8) if want_image else None
        flags = torch.empty((n,), device=dev, dtype=torch.int32)
        shadow = torch.empty((n, 4), device=dev, dtype=torch.int32) if max_depth <= 3 else None
        recs = (EpsmRecordOut * K)()
        recs[0].packed, recs[0].pflags = verts.data_ptr(), flags.data_ptr()
        if n > 1:
            recs[0].ray_stride, recs[0].packed_stride = rays.stride(0), verts.stride(0)
        recs[0].shadow = shadow.data_ptr() if shadow is not None else None
        # EPSM_TRACE_FUSE_FIRST_HIT (include/epsm_trace.h): the backward pass of the paths WITHOUT a chain -- and every path's share of
        # d / d ray.o -- done by the stage that shades the first hit; nothing of such a path is logged.  ``first_hit`` = (grad_in,
        # ParamGrads, clip).  The wavefront form's, with the gradient-only rule, without the occluder record.
        fuse = (first_hit is not None and bool(gradient_only) and shadow is None and n > 0 and self.use_wavefront(n)
                and self._backend_supports_fusion())
        fh = None
        if fuse:
            grad_in, target, clip, want_origin = first_hit
            assert grad_in.is_contiguous() and grad_in.dtype == torch.float32 and grad_in.shape[-1] >= 5
            # the paths the first-hit stage does NOT retire, in path order: the backward pass then runs its windows over these alone
            survivors = torch.empty((n,), device=dev, dtype=torch.int32) if self._backend is None else None
            survivor_count = torch.zeros((1,), device=dev, dtype=torch.int32) if self._backend is None else None
            fh = EpsmFirstHitBackward(grad_in.data_ptr(), int(grad_in.shape[1]), int(grad_in.shape[2]), int(sensor.width), float(clip),
                                      self.tri_table.data_ptr(), int(self.tri_table.shape[0]), int(target.V), target.pos.data_ptr(),
                                      target.cam_origin.data_ptr() if want_origin else None,
                                      survivors.data_ptr() if survivors is not None else None,
                                      survivor_count.data_ptr() if survivor_count is not None else None)
            recs[0].first_hit = C.addressof(fh)
        cs = sensor.c_struct()
        args = [C.byref(self.c_scene), C.byref(cs), C.c_uint32(seed & 0xFFFFFFFF), int(spp), int(max_depth), int(self.rr_depth),
                C.c_int64(lo), C.c_int64(n), K, C.c_void_p(rays.data_ptr()), None, None, None,
                C.c_void_p(film_pos.data_ptr()) if want_image else None, C.c_void_p(radiance.data_ptr()) if want_image else None,
                C.c_void_p(valid.data_ptr()) if want_image else None,
                C.c_void_p(C.addressof(recs)),
                C.c_uint32(EPSM_TRACE_SPARSE_LOG | EPSM_TRACE_PACKED_LOG | self._gradient_only_flags(gradient_only) |
                           (self._tail_flag() if self.use_wavefront(n) else 0) | (EPSM_TRACE_FUSE_FIRST_HIT if fuse else 0))]
        if self.use_wavefront(n) and n > 0:
            need = int(lib.epsm_trace_workspace_bytes(C.c_int64(n)))
            ws = self._wf_workspace.get(stream)
            if ws is None or ws.numel() < need:
                ws = self._wf_workspace[stream] = torch.empty(need, device=dev, dtype=torch.uint8)
            rc = lib.epsm_trace_paths_wavefront(*args, C.c_void_p(ws.data_ptr()), C.c_size_t(need), C.c_void_p(stream))
        else:
            rc = lib.epsm_trace_paths(*args, C.c_void_p(stream))
        if rc != 0:
            _lib.check(rc, "epsm_trace_paths") if self._backend is None else (_ for _ in ()).throw(RuntimeError(f"host tracer rc={rc}"))
        tr = PathTrace(res=sensor.width, spp=spp, ray_o=rays[:, 0:3], ray_d=rays[:, 3:6], ray_dx=rays[:, 6:9], ray_dy=rays[:, 9:12],
                       path_info=None, scatter_info=None, path_offset=lo, n_paths_total=sensor.wavefront_size(spp))
        tr.film_pos, tr.radiance, tr.valid = film_pos, radiance, valid
        tr.log = PackedLog(rays, flags, verts, shadow, self.tri_table, K)
        # the tracer has accounted for every path without a chain and for d / d ray.o: the backward kernel is called without grad_o_sum
        tr.log.first_hit_done = bool(fuse)
        if fuse and survivors is not None and want_origin:      # (the list needs a backward pass without grad_o_sum: include/epsm.h)
            tr.log.set_path_list(survivors, survivor_count)
        return tr

    def _trace(self, sensor_index: int, seed: int, spp: int, max_depth: int, K: int, lo: int, hi: int,
               want_radiance: bool = True, sparse_log: bool = False, gradient_only=None):
        from .integrators import PathTrace
        dev = self.device
        if dev.type != "cuda" and self._backend is None:
            raise _lib.EpsmError("the tracer runs on the GPU only (no CPU fallback)")
        lib = self._backend if self._backend is not None else _lib.lib()
        stream = torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else None
        sensor = self.sensors[sensor_index]
        n = hi - lo
        # One allocation per element type, carved into the per-vertex arrays with unbind(): ~10 tensor
        # constructions per tile instead of ~25 per logged vertex (the Python side of a 2^20-path tile was
        # as long as its kernel).  Arrays of consecutive vertices end up N elements apart.
        Kk = max(K, 0)
        v3 = torch.empty((5 + 10 * Kk, n, 3), device=dev, dtype=torch.float32).unbind(0)
        ray, radiance = list(v3[:4]), v3[4]
        film_pos = torch.empty((n, 2), device=dev, dtype=torch.float32)
        u8 = torch.empty((1 + 3 * Kk, n), device=dev, dtype=torch.uint8).unbind(0)
        valid = u8[0]
        recs = (EpsmRecordOut * max(1, K))()
        info, sinfo = [{"cam": ray[0]}], []
        if K > 0:
            f1 = torch.empty((3 * K, n), device=dev, dtype=torch.float32).unbind(0)
            bsdf = torch.empty((K, n), device=dev, dtype=torch.int32).unbind(0)
            tri_id = torch.empty((K, n), device=dev, dtype=torch.int32).unbind(0)
            # aux, emit (+ the occluder record of the first vertex: integrators with max_depth <= 3, epsm.py:609-620)
            want_shadow = max_depth <= 3
            quad = torch.empty((2 * K + (1 if want_shadow else 0), n, 4), device=dev, dtype=torch.int32).unbind(0)
        for k in range(K):
            t = dict(zip(("p0", "p1", "p2", "p", "n0", "n1", "n2", "normal", "hf", "light"), v3[5 + 10 * k: 15 + 10 * k]))
            t.update(zip(("b0", "b1", "eta"), f1[3 * k: 3 * k + 3]))
            t.update(zip(("active", "active_em", "ismesh"), u8[1 + 3 * k: 4 + 3 * k]))
            t["bsdf"], t["tri"], t["aux"], t["emit"] = bsdf[k], tri_id[k], quad[2 * k], quad[2 * k + 1]
            shadow = quad[2 * K] if (k == 0 and want_shadow) else None
            r = recs[k]
            for name, _ in EpsmRecordOut._fields_:
                if name in ("packed", "pflags", "ray_stride", "packed_stride", "first_hit"):
                    continue
                setattr(r, name, t[name].data_ptr() if name != "shadow" else (shadow.data_ptr() if shadow is not None else None))
            info.append({"it": k, "active": t["active"], "bsdf": t["bsdf"], "ismesh": t["ismesh"], "light": t["light"],
                         "active_em": t["active_em"], "points": [t["p0"], t["p1"], t["p2"], t["p"]],
                         "uv": [t["b0"], t["b1"]], "normal": t["normal"], "normals": [t["n0"], t["n1"], t["n2"]],
                         "eta": t["eta"], "hf": t["hf"]})
            sinfo.append({"tri": t["tri"], "aux": t["aux"] if self.alpha_slots else None, "emit": t["emit"], "shadow": shadow,
                          "table": self.tri_table})
        cs = sensor.c_struct()
        args = [C.byref(self.c_scene), C.byref(cs), C.c_uint32(seed & 0xFFFFFFFF), int(spp), int(max_depth), int(self.rr_depth),
                C.c_int64(lo), C.c_int64(n), K, C.c_void_p(ray[0].data_ptr()), C.c_void_p(ray[1].data_ptr()),
                C.c_void_p(ray[2].data_ptr()), C.c_void_p(ray[3].data_ptr()),
                C.c_void_p(film_pos.data_ptr()), C.c_void_p(radiance.data_ptr()), C.c_void_p(valid.data_ptr()),
                C.c_void_p(C.addressof(recs)),
                C.c_uint32((EPSM_TRACE_SPARSE_LOG if sparse_log else 0) | (self._gradient_only_flags(gradient_only) if K >= 1 else 0) |
                           (self._tail_flag() if self.use_wavefront(n) else 0))]
        if self.use_wavefront(n) and n > 0:
            # queues of live paths, three small kernels per bounce (include/epsm_trace.h); the workspace is scratch
            # and is kept between calls
            need = int(lib.epsm_trace_workspace_bytes(C.c_int64(n)))
            ws = self._wf_workspace.get(stream)               # per stream: launches on one stream are ordered, two streams are not
            if ws is None or ws.numel() < need:
                ws = self._wf_workspace[stream] = torch.empty(need, device=dev, dtype=torch.uint8)
            rc = lib.epsm_trace_paths_wavefront(*args, C.c_void_p(ws.data_ptr()), C.c_size_t(need), C.c_void_p(stream))
        else:
            rc = lib.epsm_trace_paths(*args, C.c_void_p(stream))
        if rc != 0:
            _lib.check(rc, "epsm_trace_paths") if self._backend is None else (_ for _ in ()).throw(RuntimeError(f"host tracer rc={rc}"))
        tr = PathTrace(res=sensor.width, spp=spp, ray_o=ray[0], ray_d=ray[1], ray_dx=ray[2], ray_dy=ray[3],
                       path_info=info, scatter_info=sinfo, path_offset=lo,
                       n_paths_total=sensor.wavefront_size(spp))
        tr.film_pos, tr.radiance, tr.valid = film_pos, radiance, valid
        return tr

    supports_packed_log = True
    supports_gradient_only = True
    supports_first_hit_fusion = True

    def _backend_supports_fusion(self) -> bool:
        return True

    def iter_traces(self, sensor=2, seed=0, spp=8, max_depth=6, max_log_depth=5, rank=0, world_size=1, sparse_log=False, first_hit=None,
                    packed_log=False, gradient_only=None):
        """Generator over this rank's tiles of the backward wavefront of ``sensors[sensor]`` (epsm.py:142-181): a tile
        is traced when the consumer asks for it, so ``render_backward`` holds ONE tile's records (~0.8 KB per path at
        K = 5) at a time whatever the size of the wavefront.  ``sparse_log``: EPSM_TRACE_SPARSE_LOG -- bounces a path
        did not reach carry only their (zero) mask fields, which is all the gradient kernels read of them; the other
        arrays are uninitialised there.  ``gradient_only`` = "manifold" / "manifold_caustic": EPSM_TRACE_GRADIENT_ONLY --
        a path is retired once nothing behind its last logged vertex can reach that variant's calc_grad (identical gradients;
        ``radiance`` is then NOT the path's radiance)."""
        si = min(sensor, len(self.sensors) - 1)
        s = self.sensors[si]
        if s.width != s.height:
            raise ValueError("the EPSM backward pass assumes a square film (epsm.py:239)")
        if s.border:
            raise ValueError("the EPSM backward pass maps path -> pixel without a sample border (epsm.py:239-246: its sensors "
                             "set sample_border = False); sample_border is for the sensor prb_reparam renders with")
        max_depth = 6 if max_depth < 0 else min(int(max_depth), 6)       # -1 = no limit; the path loop stops at 6 (epsm.py:549)
        K = min(max_log_depth, max_depth, 5)
        n_total = s.wavefront_size(spp)
        tile = self.tile_paths
        if self.use_wavefront():       # as large as the sharding allows: every rank still gets a tile, none larger than 2^24 paths
            per_rank = -(-n_total // max(1, world_size))
            tile = max(self.tile_paths, min(self.WAVEFRONT_TILE_PATHS, per_rank))
        tiles = _dist.tile_ranges(n_total, tile)
        for t in _dist.my_tiles(len(tiles), rank, world_size):
            if packed_log and K >= 1:
                yield self._trace_packed(si, seed, spp, max_depth, K, *tiles[t], gradient_only=gradient_only, first_hit=first_hit)
            else:
                yield self._trace(si, seed, spp, max_depth, K, *tiles[t], sparse_log=sparse_log, gradient_only=gradient_only)

    def trace_paths(self, *args, **kw):
        """All of this rank's tiles at once (``list(iter_traces(...))``): for small wavefronts and the tests."""
        return list(self.iter_traces(*args, **kw))

    def render_primal(self, sensor=0, seed=0, spp=0, max_depth=6, rank=None, world_size=None) -> torch.Tensor:
        """(H,W,3) image: sample_rays + path tracing + film splat / develop (epsm.py:13-76).  With more than
        one rank (default: the initialised process group) every rank traces its round-robin share of the tiles
        and the film accumulator [r,g,b,w] is summed with one all-reduce before the weight division (SURVEY 8e:
        the film is the only shared state of the primal pass), so all ranks return the same image."""
        si = min(sensor, len(self.sensors) - 1)
        s = self.sensors[si]
        spp = spp or s.spp
        n_total = s.wavefront_size(spp)
        accum = torch.zeros((s.height, s.width, 4), device=self.device, dtype=torch.float32)
        lib = self._backend if self._backend is not None else _lib.lib()
        stream = torch.cuda.current_stream(self.device).cuda_stream if self.device.type == "cuda" else None
        if rank is None or world_size is None:
            rank, world_size = _dist.world()
        tile = self.tile_paths
        if self.use_wavefront():       # as iter_traces: launches as large as the sharding allows
            tile = max(tile, min(self.WAVEFRONT_TILE_PATHS, -(-n_total // max(1, world_size))))
        tiles = _dist.tile_ranges(n_total, tile)
        for t in _dist.my_tiles(len(tiles), rank, world_size):
            lo, hi = tiles[t]
            tr = self._trace(si, seed, spp, max_depth, 0, lo, hi)
            rc = lib.epsm_film_splat(C.c_int64(hi - lo), C.c_void_p(tr.film_pos.data_ptr()), C.c_void_p(tr.radiance.data_ptr()),
                                     s.width, s.height, s.rfilter, C.c_void_p(accum.data_ptr()), C.c_void_p(stream))
            assert rc == 0, "epsm_film_splat failed"
        if world_size > 1:
            _dist.allreduce_param_grads(accum)
        img = torch.empty((s.height, s.width, 3), device=self.device, dtype=torch.float32)
        rc = lib.epsm_film_develop(s.width, s.height, C.c_void_p(accum.data_ptr()), C.c_void_p(img.data_ptr()), C.c_void_p(stream))
        assert rc == 0, "epsm_film_develop failed"
        return img
