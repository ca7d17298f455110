"""The checks of tests/golden/reference_vectors.py, written once against two small callables so that the SAME assertions run
on the host builds (tests/test_reference_vectors.py) and on the device code through the C ABI
(tests/test_gpu_reference_vectors.py):

  probe(what, rows (n, <=8) float32 [, cfg]) -> (n, 16) float32      epsm_probe (include/epsm_trace.h)
  tangent(o, d, dx, dy, gx, gy, p0, p1, p2) -> (db0, db1, dp (3))     the first-vertex tangent of ONE path
  scatter(path_info, scatter_info, out_param, out_light, out_diffuse, V) -> grad_pos (V,3) float64
"""
import ctypes as C
import math

import numpy as np
import torch

from epsm_mitsuba3_amd import scene as S
from tests.golden import reference_vectors as RV

PROBE = dict(TEA=0, PCG32=1, SAMPLER=2, MICROFACET=3, MICROFACET_SAMPLE=4, FRESNEL=5, FRESNEL_CONDUCTOR=6, RFILTER=7, PRIMARY_RAY=8,
             BSDF_SAMPLE=9, BSDF_EVAL=10)
PROBE_IN, PROBE_OUT = 8, 16


def bits_to_f32(u):
    return np.asarray(u, dtype=np.uint32).view(np.float32)


def f32_to_bits(f):
    return np.ascontiguousarray(f, dtype=np.float32).view(np.uint32)


def rows_of(cols):
    a = np.zeros((len(cols), PROBE_IN), dtype=np.float32)
    for i, c in enumerate(cols):
        a[i, :len(c)] = c
    return a


# ------------------------------------------------------------------------------------------------- python-int PCG32 / TEA
class Pcg32Py:
    """M. O'Neill's pcg32 in Python integers, seeded as Dr.Jit's PCG32::seed (pcg32_srandom_r) does."""

    def __init__(self, initstate=RV.PCG32_DEFAULT_STATE, initseq=RV.PCG32_DEFAULT_STREAM):
        M = (1 << 64) - 1
        self.state, self.inc = 0, ((initseq << 1) | 1) & M
        self.next_u32()
        self.state = (self.state + initstate) & M
        self.next_u32()

    def next_u32(self):
        M = (1 << 64) - 1
        old = self.state
        self.state = (old * RV.PCG32_MULT + self.inc) & M
        x = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
        r = old >> 59
        return ((x >> r) | (x << ((-r) & 31))) & 0xFFFFFFFF

    def next_f32(self):
        return float(bits_to_f32([(self.next_u32() >> 9) | 0x3F800000])[0] - np.float32(1.0))


def tea_float32(v1):
    return float(bits_to_f32([(int(v1) >> 9) | 0x3F800000])[0] - np.float32(1.0))


def tea_float64(v0, v1):
    x = ((int(v0) + (int(v1) << 32)) >> 12) | 0x3FF0000000000000
    return float(np.asarray([x], dtype=np.uint64).view(np.float64)[0] - 1.0)


# ------------------------------------------------------------------------------------------------- checks on the probe
def check_tea(probe):
    inp = rows_of([bits_to_f32(list(v)) for v in RV.TEA_INPUTS])
    out = f32_to_bits(probe(PROBE["TEA"], inp))
    for i, (v0, v1) in enumerate(RV.TEA_INPUTS):
        o0, o1 = int(out[i, 0]), int(out[i, 1])
        assert tea_float32(o1) == RV.TEA_FLOAT32[i], (v0, v1)                     # test_random.py:9-16, exact
        assert tea_float64(o0, o1) == RV.TEA_FLOAT64[i], (v0, v1)                 # test_random.py:20-27, exact


def check_tea_python(fn):
    """The host-side sample_tea_32 (integrators.py: derives the differential seed, util.py:505-508)."""
    for i, (v0, v1) in enumerate(RV.TEA_INPUTS):
        o0, o1 = fn(v0, v1)
        assert tea_float32(o1) == RV.TEA_FLOAT32[i] and tea_float64(o0, o1) == RV.TEA_FLOAT64[i]


def check_pcg32(probe):
    s, q = RV.PCG32_DEMO_SEED
    cases = [(s, q), (0, RV.PCG32_DEFAULT_STREAM), (RV.PCG32_DEFAULT_STATE, RV.PCG32_DEFAULT_STREAM), (0xDEADBEEF12345678, 0x1234567)]
    inp = rows_of([bits_to_f32([a & 0xFFFFFFFF, a >> 32, b & 0xFFFFFFFF, b >> 32]) for a, b in cases])
    out = probe(PROBE["PCG32"], inp)
    assert [int(x) for x in f32_to_bits(out[0, :6])] == RV.PCG32_DEMO_OUTPUT            # the published check values
    for i, (a, b) in enumerate(cases):
        py = Pcg32Py(a, b)
        assert [int(x) for x in f32_to_bits(out[i, :6])] == [py.next_u32() for _ in range(6)]
        # next_1d: (u32 >> 9 | 0x3f800000) - 1, consumed in order (test_independent.py:16-28: next_1d one draw, next_2d two)
        assert [float(x) for x in out[i, 6:12]] == [py.next_f32() for _ in range(6)]


def check_sampler_stream(probe, tea_fn):
    """The tracer's per-path stream: PCG32 seeded with (v0, v1) = sample_tea_32(seed, wavefront index), sampler.cpp:124-129."""
    cases = [(0, 0), (1, 1), (7, 123456), (0xFFFFFFFF, 0xFFFFFFFE), (1234, 16777215)]
    out = probe(PROBE["SAMPLER"], rows_of([bits_to_f32(list(c)) for c in cases]))
    for i, (seed, idx) in enumerate(cases):
        v0, v1 = tea_fn(seed, idx)
        py = Pcg32Py(v0, v1)
        assert [float(x) for x in out[i, :12]] == [py.next_f32() for _ in range(12)]


def _bsdf(distr, alpha=RV.MF_ALPHA, sample_visible=0):
    b = S.EpsmBsdf()
    b.type, b.distr, b.alpha, b.sample_visible = 2, {"beckmann": 0, "ggx": 1}[distr], alpha, sample_visible
    return b


def check_microfacet(probe):
    wi = RV.MF_WI
    mf = lambda distr, dirs: probe(PROBE["MICROFACET"], rows_of([list(m) + list(wi) for m in dirs]), _bsdf(distr)).astype(np.float64)
    # dr.allclose defaults: rtol 1e-5, atol 1e-8 (+ float32 evaluation of cos/sin of the sweep angles on our side)
    close = lambda a, b, atol=1e-8: np.allclose(a, b, rtol=2e-5, atol=atol)
    o = mf("beckmann", RV.MF_DIRS_THETA_SWEEP)
    assert close(o[:, 0], RV.BECKMANN_EVAL_THETA_SWEEP, 1e-12), o[:, 0]          # test_microfacet.py:51-58
    assert close(o[:, 1], RV.BECKMANN_PDF_THETA_SWEEP, 1e-12), o[:, 1]           # :60-67
    o = mf("beckmann", RV.MF_DIRS_PHI_SWEEP)
    assert close(o[:, 0], RV.BECKMANN_EVAL_PHI_SWEEP) and close(o[:, 1], RV.BECKMANN_PDF_PHI_SWEEP)      # :87-88
    assert np.allclose(mf("beckmann", RV.G1_DIRS_THETA_SWEEP)[:, 2], RV.BECKMANN_G1_THETA_SWEEP, rtol=1e-5, atol=1e-5)   # :110-116
    assert close(mf("beckmann", RV.G1_DIRS_PHI_SWEEP)[:, 2], RV.BECKMANN_G1_PHI_SWEEP)                                   # :129
    assert np.allclose(mf("ggx", RV.G1_DIRS_THETA_SWEEP)[:, 2], RV.GGX_G1_THETA_SWEEP, rtol=1e-5, atol=1e-5)             # :209-215
    assert close(mf("ggx", RV.G1_DIRS_PHI_SWEEP)[:, 2], RV.GGX_G1_PHI_SWEEP)                                             # :227


def check_microfacet_sample(probe):
    for distr, (m0, p0), (mh, ph) in (("beckmann", RV.BECKMANN_SAMPLE_U2_0, RV.BECKMANN_SAMPLE_U2_HALF),
                                      ("ggx", RV.GGX_SAMPLE_U2_0, RV.GGX_SAMPLE_U2_HALF)):
        for u2, m_ref, pdf_ref in ((0.0, m0, p0), (0.5, mh, ph)):
            o = probe(PROBE["MICROFACET_SAMPLE"], rows_of([[u1, u2] for u1 in RV.MF_SAMPLE_U1]), _bsdf(distr)).astype(np.float64)
            assert np.allclose(o[:, :3], m_ref, atol=5e-4), (distr, u2, o[:, :3])                    # test_microfacet.py:187, 284
            # on the alpha_u axis the anisotropic (0.1, 0.3) density is alpha_u / alpha_v of the isotropic one
            assert np.allclose(o[:, 3] / RV.MF_SAMPLE_ANISO_RATIO, pdf_ref, rtol=1e-5, atol=1e-4), (distr, u2, o[:, 3])   # :188, 285
            # and the density returned with the sample is D(m) cos(theta_m) of the distribution's own eval
            e = probe(PROBE["MICROFACET"], rows_of([list(m) + [0, 0, 1] for m in o[:, :3]]), _bsdf(distr)).astype(np.float64)
            assert np.allclose(e[:, 1], o[:, 3], rtol=2e-4)


def check_fresnel(probe):
    o = probe(PROBE["FRESNEL"], rows_of([list(i) for i, _ in RV.FRESNEL_ROWS])).astype(np.float64)
    for row, (_, ref) in zip(o, RV.FRESNEL_ROWS):
        assert np.allclose(row[:4], ref, rtol=1e-5, atol=1e-7), (row[:4], ref)                        # test_fresnel.py:8-13
    o = probe(PROBE["FRESNEL"], rows_of([list(i) for i, _, _ in RV.FRESNEL_SPOT])).astype(np.float64)
    for row, (inp, F, ct) in zip(o, RV.FRESNEL_SPOT):
        assert np.isclose(row[0], F, rtol=1e-5, atol=1e-7) and np.isclose(row[1], ct, rtol=1e-5, atol=4e-4 if ct == 0.0 else 1e-7), (row, F, ct)
        if F < 1.0:                                                                                    # :20-21, 34-35
            assert np.isclose((row[3] * math.sqrt(1 - inp[0] ** 2)) ** 2 + row[1] ** 2, 1.0, rtol=1e-5)
    c = RV.FRESNEL_MATCHED_COS
    o = probe(PROBE["FRESNEL"], rows_of([[x, 1.0] for x in c])).astype(np.float64)
    assert np.all(o[:, 0] == 0.0) and np.allclose(o[:, 1], -c, atol=5e-7)                              # :47-51
    for eta in RV.FRESNEL_CONDUCTOR_ETAS:                                                              # :54-66
        a = probe(PROBE["FRESNEL"], rows_of([[x, eta] for x in RV.FRESNEL_CONDUCTOR_COS])).astype(np.float64)[:, 0]
        b = probe(PROBE["FRESNEL_CONDUCTOR"], rows_of([[x, eta, 0.0] for x in RV.FRESNEL_CONDUCTOR_COS])).astype(np.float64)[:, 0]
        assert np.allclose(a, b, rtol=1e-5, atol=1e-6), (eta, a - b)
    th = RV.SNELL_THETA_I                                                                              # :70-77
    ct = probe(PROBE["FRESNEL"], rows_of([[math.cos(t), 1.5] for t in th])).astype(np.float64)[:, 1]
    assert np.allclose(np.sin(th) - 1.5 * np.sin(np.arccos(ct)), 0.0, atol=1e-5)


def check_rfilter(probe):
    o = probe(PROBE["RFILTER"], rows_of([[x] for x, _, _ in RV.GAUSSIAN_ROWS])).astype(np.float64)[:, 0]
    for got, (x, ref, atol) in zip(o, RV.GAUSSIAN_ROWS):
        assert abs(got - ref) <= atol, (x, got, ref)                                                   # test_rfilter.py:14-19


def check_camera(probe):
    cam = RV.CAMERA
    W, H = cam["width"], cam["height"]
    for origin in RV.CAMERA_ORIGINS:
        for direction in RV.CAMERA_DIRECTIONS:
            target = [origin[j] + direction[j] for j in range(3)]
            sensor = S.Sensor({"type": "perspective", "fov": cam["fov"], "near_clip": cam["near_clip"], "far_clip": cam["far_clip"],
                               "to_world": S.look_at(origin, target, cam["up"]),
                               "film": {"type": "hdrfilm", "width": W, "height": H, "rfilter": {"type": "box"}}})
            cs = sensor.c_struct()
            ray = lambda sx, sy: probe(PROBE["PRIMARY_RAY"], rows_of([[sx * W, sy * H]]), cs).astype(np.float64)[0]
            to_local = np.linalg.inv(S.look_at(origin, target, cam["up"]))[:3, :3]
            for sx, sy in RV.CAMERA_POS_SAMPLES:
                r = ray(sx, sy)
                o, d, dx, dy = r[0:3], r[3:6], r[6:9], r[9:12]
                inv_z = 1.0 / (to_local @ d)[2]
                assert np.allclose(o, np.asarray(origin) + cam["near_clip"] * inv_z * d, atol=1e-4)      # test_perspective.py:107-109
                assert abs(np.dot(dx - d, dy - d)) <= 1e-7                                               # :113
            c = ray(0.5, 0.5)
            assert np.allclose(c[3:6], direction, atol=1e-7)                                             # :117-118
            assert np.allclose(ray(0.5 + 1.0 / W, 0.5)[3:6], c[6:9], rtol=1e-5, atol=1e-7)               # :122-135
            assert np.allclose(ray(0.5, 0.5 + 1.0 / H)[3:6], c[9:12], rtol=1e-5, atol=1e-7)


def _plugin(kind, twosided=0, reflectance=0.5, int_ior=1.5, ext_ior=1.0):
    b = S.EpsmBsdf()
    b.type, b.twosided = {"diffuse": 0, "conductor": 1, "roughconductor": 2, "dielectric": 3}[kind], twosided
    b.reflectance[:] = [reflectance] * 3
    b.int_ior, b.ext_ior, b.alpha, b.sample_visible = int_ior, ext_ior, 0.1, 1
    b.eta[:] = [0.0] * 3; b.k[:] = [1.0] * 3
    b.alpha_slot = b.color_slot = b.texture = -1
    return b


def check_bsdf_plugins(probe):
    D = RV.DIELECTRIC
    glass = _plugin("dielectric", reflectance=D["reflectance"], int_ior=D["int_ior"], ext_ior=D["ext_ior"])
    o = probe(PROBE["BSDF_SAMPLE"], rows_of([list(wi) + [s1, 0.0, 0.0] for wi, s1, *_ in RV.DIELECTRIC_SAMPLE_ROWS]), glass).astype(np.float64)
    ob = f32_to_bits(probe(PROBE["BSDF_SAMPLE"], rows_of([list(wi) + [s1, 0.0, 0.0] for wi, s1, *_ in RV.DIELECTRIC_SAMPLE_ROWS]), glass))
    for row, bits, (wi, s1, weight, pdf, eta, wo, kind) in zip(o, ob, RV.DIELECTRIC_SAMPLE_ROWS):
        assert np.allclose(row[3:6], [weight] * 3, rtol=1e-5) and np.isclose(row[6], pdf, rtol=1e-5, atol=1e-7), (wi, s1, row)   # test_dielectric.py:40-89
        assert np.isclose(row[7], eta, rtol=1e-5) and np.allclose(row[0:3], wo, atol=1e-6) and int(bits[8]) == kind and row[9] == 1.0
    a = math.radians(RV.DIELECTRIC_SPOT_ANGLE_DEG)                                        # :141-158
    wi = [math.sin(a), 0.0, math.cos(a)]
    r = probe(PROBE["BSDF_SAMPLE"], rows_of([wi + [0.0, 0.0, 0.0], wi + [1.0, 0.0, 0.0]]), glass).astype(np.float64)
    assert np.isclose(r[0, 6], RV.DIELECTRIC_SPOT_PDF, rtol=1e-5) and np.allclose(r[0, 0:3], [-math.sin(a), 0.0, math.cos(a)], atol=1e-6)
    t = math.radians(RV.DIELECTRIC_SPOT_REFRACTED_DEG)
    assert np.isclose(r[1, 6], 1 - RV.DIELECTRIC_SPOT_PDF, rtol=1e-5) and np.allclose(r[1, 0:3], [-math.sin(t), 0.0, -math.cos(t)], atol=1e-6)
    back = probe(PROBE["BSDF_SAMPLE"], rows_of([list(r[1, 0:3]) + [1.0, 0.0, 0.0]]), glass).astype(np.float64)
    assert np.isclose(back[0, 6], 1 - RV.DIELECTRIC_SPOT_PDF, rtol=1e-5) and np.allclose(back[0, 0:3], wi, atol=1e-6)
    # diffuse (test_diffuse.py:24-35)
    wos = [[math.sin(th), 0.0, math.cos(th)] for th in RV.DIFFUSE_THETAS]
    e = probe(PROBE["BSDF_EVAL"], rows_of([[0.0, 0.0, 1.0] + wo for wo in wos]), _plugin("diffuse", reflectance=RV.DIFFUSE_DEFAULT_REFLECTANCE)).astype(np.float64)
    cos = np.array([wo[2] for wo in wos])
    live = cos > 1e-6                                  # (the last angle is pi / 2: cos = 6e-17 there, "> 0" in float64 only)
    assert np.allclose(e[live, 3], cos[live] / math.pi, rtol=1e-5) and np.allclose(e[live, 0], 0.5 * cos[live] / math.pi, rtol=1e-5)
    # twosided(diffuse) (test_twosided.py:29-45)
    two = _plugin("diffuse", twosided=1)
    e = probe(PROBE["BSDF_EVAL"], rows_of([[0, 0, 1, 0, 0, 1], [0, 0, 1, 0, 0, -1]]), two).astype(np.float64)
    assert np.isclose(e[0, 3], 1 / math.pi, rtol=1e-5) and e[1, 3] == 0.0


ALL_PROBE_CHECKS = (check_tea, check_pcg32, check_microfacet, check_microfacet_sample, check_fresnel, check_rfilter, check_camera,
                    check_bsdf_plugins)


# ------------------------------------------------------------------------------------------------- tangent and scatter
def rect_triangle(which):
    V = RV.RECT_VERTICES
    return [V[j] for j in which]


def check_tangent(tangent, rel=1e-5):
    """d p / d ray.d.x = (10, 0, 0) (test_mesh.py:414-418) and -- the plane being perpendicular to the ray, so that moving
    d by delta moves the hit by t delta, which :399-402 against :414-418 states -- d uv / d ray.d.x = t x (0.5, 0) = (5, 0)
    (:404-407), through the tangent of the product: grad_d = (d_x - d) gx + (d_y - d) gy (epsm.py:255)."""
    o, d = (np.asarray(v, dtype=np.float64) for v in RV.TANGENT_RAY)
    p0, p1, p2 = rect_triangle(RV.RECT_LOWER)
    uv = [RV.RECT_TEXCOORDS[j] for j in RV.RECT_LOWER]
    # (the second row by linearity in gx, the third by the x <-> y symmetry of the rectangle)
    for (gx, gy), dp_ref, duv_ref in (((1.0, 0.0), (10.0, 0.0, 0.0), (5.0, 0.0)), ((0.25, 0.0), (2.5, 0.0, 0.0), (1.25, 0.0)),
                                      ((0.0, 1.0), (0.0, 10.0, 0.0), (0.0, 5.0))):
        dx = d + np.array([1.0, 0.0, 0.0]); dy = d + np.array([0.0, 1.0, 0.0])
        db0, db1, dp = tangent(o, d, dx, dy, gx, gy, p0, p1, p2)
        assert np.allclose(dp, dp_ref, rtol=rel, atol=1e-5), (dp, dp_ref)
        duv = db0 * uv[0] + db1 * uv[1] - (db0 + db1) * uv[2]
        assert np.allclose(duv, duv_ref, rtol=rel, atol=1e-5), (duv, duv_ref)


def rect_path_info(dtype=torch.float32):
    """One path whose first vertex is the reference's hit of the rectangle near its 4th vertex (test_mesh.py:560-562), logged
    as the tracer would log it for a mesh WITHOUT vertex normals (mesh.cpp:811-816: n0 = n1 = n2 = si.n)."""
    o, d = (np.asarray(v, dtype=np.float64) for v in RV.SCATTER_RAY)
    p0, p1, p2 = rect_triangle(RV.RECT_UPPER)
    e1, e2 = p1 - p0, p2 - p0
    pv = np.cross(d, e2); inv = 1.0 / e1.dot(pv); tv = o - p0
    u = tv.dot(pv) * inv; v = d.dot(np.cross(tv, e1)) * inv
    b1, b2 = u, v; b0 = 1.0 - b1 - b2
    t = lambda a: torch.tensor(np.asarray(a, dtype=np.float64).reshape(1, -1), dtype=dtype)
    s = lambda a: torch.tensor([a], dtype=dtype)
    nz = [0.0, 0.0, 1.0]
    info = [{"cam": t(o)},
            {"it": 0, "active": torch.ones(1, dtype=torch.bool), "bsdf": torch.tensor([0x8 | 0x8000], dtype=torch.int32),
             "ismesh": s(1.0), "light": t([0.0, 0.0, -5.0]), "active_em": torch.ones(1, dtype=torch.bool),
             "points": [t(p0), t(p1), t(p2), t(p0 * b0 + p1 * b1 + p2 * b2)], "uv": [s(b0), s(b1)], "normal": t(nz),
             "normals": [t(nz), t(nz), t(nz)], "eta": s(1.0), "hf": t([0.0, 0.0, 1.0])}]
    vidx = torch.tensor([list(RV.RECT_UPPER)], dtype=torch.int64)
    return info, [{"vidx": vidx, "mode": 4}], (b0, b1, b2)                  # mode: positions attached, no vertex normals


def check_scatter(scatter):
    """Gather adjoints of si.p, si.n and si.sh_frame.n into vertex_positions (test_mesh.py:576-581, 611-639) through the
    scatter's `si_follow.p . diffuse_grad` (epsm.py:561-562) and `sh_frame.n . path_grad[5 it + 3]` (epsm.py:645) terms."""
    pi, si, _ = rect_path_info()
    for name, ref in RV.SCATTER_ROWS:
        op, ol, od = torch.zeros((5, 1, 3)), torch.zeros((1, 1, 3)), torch.zeros((1, 1, 3))
        what, comp = name.rsplit(".", 1)
        c = "xyz".index(comp)
        if what == "p":
            od[0, 0, c] = 1.0
        else:                                    # si.n and si.sh_frame.n of a mesh without vertex normals are the face normal
            op[3, 0, c] = 1.0
        gp = scatter(pi, si, op, ol, od, 4)
        assert np.allclose(np.asarray(gp, dtype=np.float64).reshape(-1), ref, atol=RV.SCATTER_ATOL), (name, gp)
