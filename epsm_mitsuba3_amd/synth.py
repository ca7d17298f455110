"""Seeded synthetic path records in the reference's ``path_info`` format.

Scene assets of the reference are not in its repository (README.md:26), so the
benchmark and the parity tests run on synthetic records whose *shape* follows
the record logged by ``EPSMIntegrator.sample_path`` (epsm.py:547, 648-654) and
whose *values* follow the generator specified in SURVEY.md section 8(d).

``path_info[0] = {"cam": (N,3)}``; ``path_info[k]`` for k = 1..K holds
``active (N) bool, bsdf (N) int32 flags, ismesh (N) f32, light (N,3),
active_em (N) bool, points [p0,p1,p2,p] 4x(N,3), uv [b0,b1] 2x(N),
normal (N,3), normals [n0,n1,n2] 3x(N,3), eta (N), hf (N,3)``.
"""
from __future__ import annotations

import torch

# include/mitsuba/render/bsdf.h:40-101
BSDF_NULL = 0x1
BSDF_DIFFUSE_REFLECTION = 0x2
BSDF_GLOSSY_REFLECTION = 0x8
BSDF_DELTA_REFLECTION = 0x20
BSDF_DELTA_TRANSMISSION = 0x40
BSDF_FRONT_SIDE = 0x8000
BSDF_BACK_SIDE = 0x10000
BSDF_NON_SYMMETRIC = 0x4000
BSDF_DIFFUSE = 0x2 | 0x4

FLAGS_DIFFUSE = BSDF_DIFFUSE_REFLECTION | BSDF_FRONT_SIDE
FLAGS_ROUGHCONDUCTOR = BSDF_GLOSSY_REFLECTION | BSDF_FRONT_SIDE
FLAGS_DIELECTRIC = (BSDF_DELTA_REFLECTION | BSDF_DELTA_TRANSMISSION | BSDF_FRONT_SIDE
                    | BSDF_BACK_SIDE | BSDF_NON_SYMMETRIC)
FLAGS_NULL = BSDF_NULL | BSDF_FRONT_SIDE | BSDF_BACK_SIDE

PROFILES = ("bathroom", "caustic", "pool", "specular", "mixed")


def _u(gen, shape, lo, hi, device, dtype):
    return torch.rand(shape, generator=gen, device=device, dtype=dtype) * (hi - lo) + lo


def synth_path_info(n_paths: int, n_vertices: int, seed: int = 0, device="cpu",
                    profile: str = "bathroom", dtype=torch.float32,
                    p_terminate: float = 0.1, p_no_light: float = 0.1,
                    p_not_mesh: float = 0.02, tangent_scale: float = 1e-3):
    """Returns ``(path_info, dlduv, dldp)``.

    ``dlduv`` has the reference shape ``(N, 1, 2L)`` with L = K+1 and only the
    first two columns non-zero (epsm.py:256, 268-269); ``dldp`` is ``(N, 3)``.

    Profiles (diffuse-vertex placement):
      bathroom  vertex 1 specular w.p. 0.7; P(diffuse at k)=0.6 for k>=2;
                reflection (eta=1) with 20 % dielectric refraction vertices.
      caustic   vertex 1 diffuse, then 1..3 specular vertices, then diffuse.
      pool      as caustic but every specular vertex refracts (eta 1.33 / 1/1.33),
                chain length uniform in 1..K-1 ("deep specular chains").
      specular  no diffuse vertex at all (full-length chains, worst case work).
      mixed     each of the above on a quarter of the paths, plus Null vertices.
    """
    if profile not in PROFILES:
        raise ValueError(f"unknown profile {profile!r}")
    N, K = int(n_paths), int(n_vertices)
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234567 + 7919 * int(seed))
    f = dict(device=dev, dtype=dtype)

    cam = torch.tensor([0.0, 0.0, 5.0], **f).expand(N, 3).contiguous()
    info = [{"cam": cam}]

    # --- which vertices are diffuse -------------------------------------
    r = torch.rand((N, K), generator=gen, device=dev)
    if profile == "mixed":
        sel = torch.randint(0, 4, (N,), generator=gen, device=dev)
    else:
        sel = torch.full((N,), {"bathroom": 0, "caustic": 1, "pool": 2, "specular": 3}[profile],
                         device=dev, dtype=torch.int64)
    kk = torch.arange(1, K + 1, device=dev)[None, :]
    diff_bath = torch.where(kk == 1, r < 0.3, r < 0.6)
    chain = torch.randint(1, max(2, min(4, K)), (N, 1), generator=gen, device=dev)
    diff_caus = (kk == 1) | (kk >= chain + 2)
    chain_p = torch.randint(1, max(2, K), (N, 1), generator=gen, device=dev)
    diff_pool = (kk == 1) | (kk >= chain_p + 2)
    diff_spec = torch.zeros((N, K), dtype=torch.bool, device=dev)
    s = sel[:, None]
    is_diffuse = torch.where(s == 0, diff_bath, torch.where(s == 1, diff_caus,
                             torch.where(s == 2, diff_pool, diff_spec)))

    alive = torch.ones((N,), dtype=torch.bool, device=dev)
    for k in range(1, K + 1):
        # triangle centres zig-zag so that consecutive vertices are >= 1.5 apart
        centre = torch.tensor([1.6 * ((k % 2) * 2 - 1) * 0.5, 0.35 * k, 2.5 - 2.0 * (k % 2)], **f)
        p = [centre + _u(gen, (N, 3), -0.5, 0.5, dev, dtype) for _ in range(3)]
        b0 = _u(gen, (N,), 0.0, 0.5, dev, dtype)
        b1 = _u(gen, (N,), 0.0, 0.5, dev, dtype)
        base_n = torch.tensor([0.1, 0.2, 1.0], **f)
        nrm = []
        for _ in range(3):
            v = base_n + 0.2 * _u(gen, (N, 3), -0.5, 0.5, dev, dtype)
            nrm.append(v / torch.linalg.norm(v, dim=-1, keepdim=True))
        pos = p[0] * b0[:, None] + p[1] * b1[:, None] + p[2] * (1 - b0 - b1)[:, None]
        nint = nrm[0] * b0[:, None] + nrm[1] * b1[:, None] + nrm[2] * (1 - b0 - b1)[:, None]
        nint = nint / torch.linalg.norm(nint, dim=-1, keepdim=True)

        refr = torch.rand((N,), generator=gen, device=dev)
        enter = torch.rand((N,), generator=gen, device=dev) < 0.5
        eta_glass = torch.where(enter, torch.tensor(1.5, **f), torch.tensor(1.0 / 1.5, **f))
        eta_water = torch.where(enter, torch.tensor(1.33, **f), torch.tensor(1.0 / 1.33, **f))
        is_refr = torch.where(sel == 2, torch.ones_like(enter), refr < 0.2)
        eta = torch.where(is_refr, torch.where(sel == 2, eta_water, eta_glass), torch.tensor(1.0, **f))
        dflag = is_diffuse[:, k - 1]
        eta = torch.where(dflag, torch.tensor(1.0, **f), eta)
        flags = torch.where(dflag, torch.tensor(FLAGS_DIFFUSE, device=dev),
                            torch.where(is_refr, torch.tensor(FLAGS_DIELECTRIC, device=dev),
                                        torch.tensor(FLAGS_ROUGHCONDUCTOR, device=dev)))
        if profile == "mixed":
            null_v = torch.rand((N,), generator=gen, device=dev) < 0.03
            flags = torch.where(null_v & ~dflag, torch.tensor(FLAGS_NULL, device=dev), flags)
        hf = torch.cat([0.05 * _u(gen, (N, 2), -0.5, 0.5, dev, dtype),
                        torch.ones((N, 1), **f)], dim=-1)
        # only roughconductor exports hf (roughconductor.cpp:255); zero elsewhere
        hf = torch.where((flags == FLAGS_ROUGHCONDUCTOR)[:, None], hf, torch.zeros_like(hf))
        light = torch.tensor([0.0, 4.0, 4.0], **f) + _u(gen, (N, 3), 0.0, 1.0, dev, dtype)

        if k > 1:
            alive = alive & (torch.rand((N,), generator=gen, device=dev) >= p_terminate)
        active_em = alive & (torch.rand((N,), generator=gen, device=dev) >= p_no_light)
        ismesh = (torch.rand((N,), generator=gen, device=dev) >= p_not_mesh).to(dtype)

        info.append({
            "it": k - 1,
            "active": alive.clone(),
            "bsdf": flags.to(torch.int32),
            "ismesh": ismesh,
            "light": light,
            "active_em": active_em,
            "points": [p[0], p[1], p[2], pos],
            "uv": [b0, b1],
            "normal": nint,
            "normals": nrm,
            "eta": eta,
            "hf": hf,
        })

    L = K + 1
    dlduv = torch.zeros((N, 1, 2 * L), **f)
    dlduv[:, 0, :2] = torch.randn((N, 2), generator=gen, device=dev, dtype=dtype) * tangent_scale
    dldp = torch.randn((N, 3), generator=gen, device=dev, dtype=dtype) * tangent_scale
    return info, dlduv, dldp


def path_info_to(path_info, device=None, dtype=None):
    """Deep copy of a ``path_info`` list onto another device / float dtype."""
    out = []
    for rec in path_info:
        r = {}
        for k, v in rec.items():
            def conv(t):
                t = t.detach().clone()
                if dtype is not None and t.is_floating_point():
                    t = t.to(dtype)
                if device is not None:
                    t = t.to(device)
                return t
            if isinstance(v, (list, tuple)):
                r[k] = [conv(x) for x in v]
            elif isinstance(v, torch.Tensor):
                r[k] = conv(v)
            else:
                r[k] = v
        out.append(r)
    return out


# ---------------------------------------------------------------------------
# camera rays, first-hit triangles and parameter addressing (tangent / scatter)
# ---------------------------------------------------------------------------
def synth_camera_rays(res: int, spp: int, seed: int = 0, device="cpu", fov_deg: float = 40.0,
                      dtype=torch.float32, lo: int = 0, hi=None):
    """Primary rays of a pinhole camera at the origin looking down -z with the
    one-pixel offset directions of ``sample_ray_differential``
    (src/sensors/perspective.cpp:238-279).  Paths are ordered (pixel, sample),
    pixels row-major, as the reference reshapes them (epsm.py:250)."""
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(424242 + int(seed) + 1000003 * int(lo))
    hi = res * res * spp if hi is None else int(hi)
    N = hi - int(lo)
    pix = torch.arange(int(lo), hi, device=dev) // spp
    py, px = (pix // res).to(dtype), (pix % res).to(dtype)
    jit = torch.rand((N, 2), generator=gen, device=dev, dtype=dtype)
    import math
    scale = 2.0 * math.tan(math.radians(fov_deg) / 2.0) / res

    def direction(sx, sy):
        v = torch.stack([(sx - res / 2) * scale, -(sy - res / 2) * scale, -torch.ones_like(sx)], dim=-1)
        return v / torch.linalg.norm(v, dim=-1, keepdim=True)

    sx, sy = px + jit[:, 0], py + jit[:, 1]
    d = direction(sx, sy)
    dx = direction(sx + 1, sy)
    dy = direction(sx, sy + 1)
    o = torch.zeros((N, 3), device=dev, dtype=dtype)
    return o, d, dx, dy


def synth_first_hit_triangles(o, d, seed: int = 0):
    """Triangles ``p0,p1,p2`` that the rays hit strictly inside, plus the reference
    barycentrics ``(b0,b1)`` of the hit (b1 = u, b0 = 1-u-v; mesh.cpp:698-700)."""
    dev, dtype, N = d.device, d.dtype, d.shape[0]
    gen = torch.Generator(device=dev)
    gen.manual_seed(777 + int(seed))
    t = _u(gen, (N, 1), 2.0, 4.0, dev, dtype)
    hit = o + t * d
    u = _u(gen, (N, 1), 0.1, 0.4, dev, dtype)
    v = _u(gen, (N, 1), 0.1, 0.4, dev, dtype)
    # edges roughly perpendicular to the viewing direction, size ~0.3
    a = torch.tensor([1.0, 0.2, 0.1], device=dev, dtype=dtype) + 0.3 * _u(gen, (N, 3), -0.5, 0.5, dev, dtype)
    b = torch.tensor([0.1, 1.0, 0.3], device=dev, dtype=dtype) + 0.3 * _u(gen, (N, 3), -0.5, 0.5, dev, dtype)
    e1, e2 = 0.3 * a, 0.3 * b
    p0 = hit - u * e1 - v * e2
    return p0, p0 + e1, p0 + e2, (1 - u - v)[:, 0], u[:, 0]


def synth_triangle_table(n_scene_vertices: int, device="cpu", n_emitter_tris: int = 32, n_bsdfs: int = 4) -> torch.Tensor:
    """The synthetic scene's triangle table ``(T,4) int32 [v0, v1, v2, mode]`` (include/epsm.h): V surface triangles
    -- triangle t uses vertex rows (t, t+1, t+G) mod V, G = sqrt(2V) -- followed by ``n_emitter_tris`` emitter triangles
    on the last 3 * n_emitter_tris vertex rows.  Modes per triangle (a fixed pseudo-random function of V): 60 % smooth
    and attached, 25 % flat meshes, 10 % with flipped normals, 5 % detached; bits 8.. of the mode word carry the alpha
    slot of the triangle's BSDF + 1 (uniform over none, 0..n_bsdfs-1), as the packed log wants it (include/epsm.h).
    Depends only on its arguments, so every tile / rank / slab of a scene shares it."""
    import math
    V = int(n_scene_vertices)
    dev = torch.device(device)
    G = max(2, int(math.sqrt(2 * V)))
    gen = torch.Generator(device="cpu").manual_seed(5150 + V)
    r = torch.rand((V,), generator=gen)
    mode = torch.full((V,), 4 | 8 | 1, dtype=torch.int64)                               # attached, vertex normals
    mode = torch.where(r < 0.25, torch.tensor(4), mode)                                 # flat mesh
    mode = torch.where((r >= 0.25) & (r < 0.35), torch.tensor(4 | 8 | 1 | 2), mode)     # flipped normals
    mode = torch.where((r >= 0.35) & (r < 0.40), torch.tensor(1), mode)                 # detached
    slot = torch.randint(-1, max(0, int(n_bsdfs)), (V,), generator=gen) if n_bsdfs > 0 else torch.full((V,), -1)
    mode = mode | ((slot + 1) << 8)
    t = torch.arange(V)
    rows = [torch.stack([t, (t + 1) % V, (t + G) % V, mode], dim=1)]
    if n_emitter_tris > 0 and V >= 3 * n_emitter_tris:
        e = V - 3 * n_emitter_tris + 3 * torch.arange(n_emitter_tris)
        rows.append(torch.stack([e, e + 1, e + 2, torch.full_like(e, 4)], dim=1))
    return torch.cat(rows, dim=0).to(torch.int32).to(dev).contiguous()


def synth_scatter_info(n_paths: int, n_vertices: int, n_scene_vertices: int, seed: int = 0, device="cpu",
                       n_bsdfs: int = 4, coherent: bool = True, dtype=torch.float32, n_emitter_tris: int = 32,
                       res: int = 0, spp: int = 1, path_offset: int = 0, shadow: bool = False, table=None):
    """Per-vertex parameter addressing (``EpsmScatterRecord`` fields, packed) of a synthetic trace over the scene of
    ``synth_triangle_table``; every returned dict also carries that table under ``"table"``.

    Coherence model (``coherent=True``; documented in DESIGN.md 7): the scene's surface triangles are laid out on a
    virtual G x G grid, G = sqrt(2V); the first hit of a path is the triangle under its pixel (``res`` x ``res`` film
    mapped onto the grid, so a triangle covers (res/G)^2 pixels and all ``spp`` samples of a pixel share it); every
    further bounce jitters the grid cell by +-2^k cells, i.e. neighbouring paths keep hitting nearby triangles but
    spread out with depth.  With ``res = 0`` paths are grouped 16 at a time instead of by pixel.  ``coherent=False``
    draws every triangle uniformly.  3 % of the hits carry no triangle (EPSM_NO_INDEX).
    Emitter samples land on the small emitter mesh at the end of the table (half of them nowhere), as area lights are
    a handful of triangles in the reference's scenes -- every wave then adds to the same few rows.
    ``shadow=True`` adds the occluder record of the first vertex (epsm.py:609-620, integrators with max_depth <= 3): a
    triangle near the first hit's cell, dis in [0, 0.9) with 30 % zeros, 5 % without a triangle."""
    N, K, V = int(n_paths), int(n_vertices), int(n_scene_vertices)
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(99991 + int(seed))
    if table is None:
        table = synth_triangle_table(V, dev, n_emitter_tris, n_bsdfs)
    has_emitters = table.shape[0] > V
    idx = torch.arange(N, device=dev) + int(path_offset)
    import math
    G = max(2, int(math.sqrt(2 * V)))
    if res > 0:
        pix = idx // max(1, spp)
        cx = ((pix % res) * G) // res
        cy = (((pix // res) % res) * G) // res
    else:
        grp = idx // 16
        cx, cy = grp % G, (grp // G) % G
    del idx
    none = torch.tensor(-1, device=dev, dtype=torch.int32)
    bits = lambda t: t.to(torch.float32).contiguous().view(torch.int32)
    info = []
    for k in range(1, K + 1):
        if coherent:
            if k > 1:
                spread = 2 ** k
                cx = (cx + torch.randint(-spread, spread + 1, (N,), generator=gen, device=dev)) % G
                cy = (cy + torch.randint(-spread, spread + 1, (N,), generator=gen, device=dev)) % G
            base = ((cy * G + cx) + k * 7) % V
        else:
            base = torch.randint(0, V, (N,), generator=gen, device=dev)
        r = torch.rand((N,), generator=gen, device=dev)
        tri = torch.where((r >= 0.40) & (r < 0.43), none, base.to(torch.int32))
        if has_emitters:
            etri = (V + torch.randint(0, n_emitter_tris, (N,), generator=gen, device=dev)).to(torch.int32)
        else:
            etri = torch.randint(0, V, (N,), generator=gen, device=dev).to(torch.int32)
        etri = torch.where(torch.rand((N,), generator=gen, device=dev) < 0.5, etri, none)
        rec = {"tri": tri, "table": table}
        if shadow and k == 1:
            stri = ((base + 11 + torch.randint(0, 3, (N,), generator=gen, device=dev)) % V).to(torch.int32)
            stri = torch.where(torch.rand((N,), generator=gen, device=dev) < 0.05, none, stri)
            sdis = _u(gen, (N,), 0.0, 0.9, dev, dtype)
            sdis = torch.where(torch.rand((N,), generator=gen, device=dev) < 0.3, torch.zeros_like(sdis), sdis)
            rec["shadow"] = torch.stack([stri, bits(_u(gen, (N,), 0.0, 0.5, dev, dtype)), bits(_u(gen, (N,), 0.0, 0.5, dev, dtype)),
                                         bits(sdis)], dim=1)
        # the alpha slot belongs to the triangle's BSDF (mode word of its table row, bits 8..); none without a triangle
        bsdf_id = torch.where(tri >= 0, (table[:, 3][tri.clamp_min(0).long()] >> 8) - 1, none).to(torch.int32)
        rec["aux"] = torch.cat([bsdf_id[:, None], bits(_u(gen, (N, 3), -1.0, 1.0, dev, dtype))], dim=1)
        del bsdf_id
        rec["emit"] = torch.stack([etri, bits(_u(gen, (N,), 0.0, 0.5, dev, dtype)), bits(_u(gen, (N,), 0.0, 0.5, dev, dtype)),
                                   bits(_u(gen, (N,), 0.0, 2.0, dev, dtype))], dim=1)
        del etri, tri, r, base
        info.append(rec)
    return info
