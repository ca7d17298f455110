"""`prb_reparam`: gradients of vertex positions / normals through visibility (csrc/epsm_trace_reparam.h)
on the HOST build of the per-path code (tests/host_harness), by the reference's own recipe for this integrator --
src/integrators/tests/test_ad_integrators.py:833-871: the backward gradient against finite differences of the primal image
(its thresholds: 10 % .. 35 % of the gradient, per config) -- plus closed forms where there are any.  The GPU twin with the
reference's sample counts is tests/test_gpu_reparam.py."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

import epsm_mitsuba3_amd as epsm
from _reparam_scenes import CONFIGS, build, fd_check, fd_check_normals, rect
from _scenes import host_tracer, on_host, sensor
from epsm_mitsuba3_amd import scene as S


def rel(got, fd):
    g, f = float(np.mean(got)), float(np.mean(fd))
    return abs(g - f) / max(abs(f), 1e-3), g, f


def test_warp_field_of_a_plane_against_its_closed_form():
    """Only the ray origin moves (velocity u) in front of a plane at distance H with unit normal m towards it: the field is
    V(w) = -(u - w (w.u)) (w.m) / H and its divergence on the sphere -(m.u - 3 (w.m)(w.u)) / H.  The estimate of V is a
    weighted mean of exact values; that of the divergence is the self-normalised one of reparam.py:213-215, whose bias
    goes as 1 / rays (-4 % at 16, -2.5 % at 64)."""
    v, f = rect(5.0, (0, 0, 2.0))
    d = {"type": "scene", "cam": sensor([0, 0, 4], [0, 0, 0], res=8),
         "light": {"type": "mesh", "vertices": v, "faces": f[:, ::-1], "face_normals": True,
                   "emitter": {"type": "area", "radiance": {"type": "rgb", "value": [1.0, 1.0, 1.0]}}}}
    sc = on_host(S.Scene.from_dict(d, device="cpu"))
    lib = host_tracer()
    o = np.array([0.3, 0.1, 0.0], np.float32)
    m, H = np.array([0, 0, 1.0], np.float32), 2.0
    out = (C.c_float * 5)()
    for w in ([0, 0, 1.0], [0.5, 0.2, 0.84]):
        w = np.array(w, np.float32); w /= np.linalg.norm(w)
        for u in ([0, 0, 1.0], [1.0, 0, 0]):
            u = np.array(u, np.float32)
            acc, n = np.zeros(5), 1500
            for s in range(n):
                lib.epsm_debug_warp(C.byref(sc.c_scene), o.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p),
                                    u.ctypes.data_as(C.c_void_p), 64, C.c_float(1e5), C.c_float(3.0), C.c_uint32(0), C.c_uint32(s), out)
                acc += np.array(out[:])
            acc /= n
            V = -(u - w * (w @ u)) * (w @ m) / H
            div = -(m @ u - 3 * (w @ m) * (w @ u)) / H
            assert np.allclose(acc[:3], V, atol=2e-4), (w, u, acc, V)
            assert abs(acc[3] - div) <= 0.05 * abs(div) + 1e-3, (w, u, acc[3], div)


def test_antithetic_pairs_mirror_the_sample():
    """`reparam_antithetic` (reparam.py:82-84, 189-196): rays 2m and 2m + 1 share a sample, the even one mirrored about the
    ray.  On the plane: the field's value stays exact (a weighted mean of exact values); for a ray along the normal and a
    motion across it everything odd cancels PER WARP (dZ = 0, divergence 0 to fp32 rounding, where independent rays leave 25 x
    the noise);
    and the self-normalised divergence estimate of reparam.py:213-215 stays consistent -- its bias, larger with pairs than
    without (a property of that estimator: an independent numpy restatement of it gives the same +20..+100 % at 16 rays
    depending on the boundary term), falls as 1 / rays."""
    v, f = rect(5.0, (0, 0, 2.0))
    d = {"type": "scene", "cam": sensor([0, 0, 4], [0, 0, 0], res=8),
         "light": {"type": "mesh", "vertices": v, "faces": f[:, ::-1], "face_normals": True,
                   "emitter": {"type": "area", "radiance": {"type": "rgb", "value": [1.0, 1.0, 1.0]}}}}
    sc = on_host(S.Scene.from_dict(d, device="cpu"))
    lib = host_tracer()
    o = np.array([0.3, 0.1, 0.0], np.float32)
    m, H = np.array([0, 0, 1.0], np.float32), 2.0
    out = (C.c_float * 5)()
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)

    def runs(w, u, rays, flags, n):
        vals = []
        for s in range(n):
            lib.epsm_debug_warp(C.byref(sc.c_scene), ptr(o), ptr(w), ptr(u), rays, C.c_float(1e5), C.c_float(3.0), C.c_uint32(flags),
                                C.c_uint32(s), out)
            vals.append(np.array(out[:]))
        return np.array(vals)

    wn = np.array([0, 0, 1.0], np.float32)
    wo = np.array([0.5, 0.2, 0.84], np.float32); wo /= np.linalg.norm(wo)
    ux, uz = np.array([1.0, 0, 0], np.float32), np.array([0, 0, 1.0], np.float32)
    for w in (wn, wo):
        for u in (ux, uz):
            V = -(u - w * (w @ u)) * (w @ m) / H
            assert np.allclose(runs(w, u, 16, 1, 200)[:, :3].mean(0), V, atol=3e-4), (w, u)
    plain, paired = runs(wn, ux, 16, 0, 100), runs(wn, ux, 16, 1, 100)
    assert paired[:, 3].std() < 0.1 * plain[:, 3].std() and plain[:, 3].std() > 1e-4, (paired[:, 3].std(), plain[:, 3].std())
    div = -(m @ uz - 3 * (wo @ m) * (wo @ uz)) / H
    b16 = runs(wo, uz, 16, 1, 800)[:, 3].mean() - div
    b64 = runs(wo, uz, 64, 1, 800)[:, 3].mean() - div
    assert abs(b64) < 0.5 * abs(b16) and abs(b64) < 0.12 * abs(div), (b16, b64, div)


def test_antithetic_backward_pass_matches_the_closed_form_of_a_moving_emitter():
    """The whole pass with mirrored pairs on the configuration of test_emitter_moving_inside_the_image...: same expectation."""
    name, res, spp = "emitter_in_view", 32, 256
    sc = build(name, 0.0, res, spp)
    sc.attach("light", positions=True)
    integ = epsm.load_dict({"type": "prb_reparam", "max_depth": 2, "reparam_rays": 32, "reparam_antithetic": True})
    g = torch.ones((res, res, 3)) * (0.5 + torch.arange(res, dtype=torch.float32) / res)[None, :, None]
    got = []
    for seed in range(2):
        p = sc.param_grads()
        integ.render_backward(sc, p, g, sensor=0, seed=seed, spp=spp)
        got.append(float(p.mesh_pos("light")[:, 0].sum()))
    ppu = res / (2 * 4 * math.tan(math.radians(28.8415 / 2)))
    want = 3 * ppu * ppu * ppu / res
    assert abs(np.mean(got) - want) < 0.08 * want, (got, want)


def test_point_light_receiver_matches_finite_differences_to_a_percent():
    """No stochastic part beyond the camera ray: primary warp, film adjoint (position + determinant), the triangle's own
    vertices and the attached 1 / r^2 of point.cpp:154-164 -- 0.1 % here."""
    for weights in ("ramp", "ones"):
        r, g, f = rel(*fd_check("receiver_point_light", spp=64, rays=16, weights=weights)[:2])
        assert r < 0.01, (weights, g, f)


def test_emitter_moving_inside_the_image_matches_the_closed_form():
    """A 1 x 1 emitter wholly in view, radiance 1, translated along x under the weights 0.5 + x / width: its image shifts, so
    d loss / d theta = 3 channels * pixels_per_unit * (area in pixels) / width."""
    name, res, spp = "emitter_in_view", 32, 256
    sc = build(name, 0.0, res, spp)
    sc.attach("light", positions=True)
    integ = epsm.load_dict({"type": "prb_reparam", "max_depth": 2, "reparam_rays": 32})
    g = torch.ones((res, res, 3)) * (0.5 + torch.arange(res, dtype=torch.float32) / res)[None, :, None]
    got = []
    for seed in range(2):
        p = sc.param_grads()
        integ.render_backward(sc, p, g, sensor=0, seed=seed, spp=spp)
        got.append(float(p.mesh_pos("light")[:, 0].sum()))
        assert abs(float(p.mesh_pos("light")[:, 1].sum())) < 0.15 * abs(got[-1])      # nothing changes along y
    ppu = res / (2 * 4 * math.tan(math.radians(28.8415 / 2)))
    want = 3 * ppu * ppu * ppu / res
    assert abs(np.mean(got) - want) < 0.08 * want, (got, want)


@pytest.mark.parametrize("name,rays,spp,tol", [
    ("receiver_along_normal", 32, 128, 0.08),        # no discontinuity in view: the emitter ray's warp + divergence
    ("corner_along_normal", 32, 128, 0.08),          # + the neighbours' BSDFs (`extra`, prb_reparam.py:515-542)
    ("rectangle_emitter_on_black", 32, 128, 0.15),   # the reference's thresholds for these two: 0.2 and 0.15
    ("sphere_emitter_on_black", 32, 128, 0.15),
    ("scale_sphere_emitter_on_black", 32, 128, 0.1),  # :438-460, its threshold
    ("diffuse_rectangle_constant", 32, 128, 0.15),    # :341-363 under the `constant` environment emitter (its threshold: 0.25)
    ("shadow_receiver_constant", 32, 256, 0.2),       # :482-520 (0.25)
    ("textured_plane_constant", 32, 128, 0.1),        # :523-550, its mean threshold (a synthetic texture for its museum.exr)
    ("textured_plane_fills_the_view", 16, 256, 0.08), # no silhouette in view: the derivative of the texture lookup alone
])
def test_smooth_and_silhouette_configs_match_finite_differences(name, rays, spp, tol):
    r, g, f = rel(*fd_check(name, spp=spp, rays=rays, seeds=1, fd_spp_mult=4)[:2])
    assert r < tol, (name, g, f)


@pytest.mark.parametrize("name,tol", [("occluder_area_light", 0.35), ("diffuse_sphere_area_light", 0.35), ("sphere_on_glossy_floor", 0.35)])
def test_shadow_and_indirect_configs_have_the_sign_and_size_of_finite_differences(name, tol):
    """The harmonic weights make these estimates heavy-tailed: at the few hundred samples per pixel a CPU test affords the
    mean is still 10-30 % short; at the reference's 2048 samples they are within its thresholds (tests/test_gpu_reparam.py,
    tools/try_reparam_fd.py: occluder 14 %, sphere 7 %)."""
    r, g, f = rel(*fd_check(name, spp=256, rays=64, seeds=1, fd_eps=5e-3, fd_spp_mult=4)[:2])
    assert g * f > 0 and r < tol, (name, g, f)


def test_vertex_normals_receive_their_gradient():
    got, fd = fd_check_normals("diffuse_sphere_area_light", spp=128, rays=16)
    # the shading rim (wi.z <= 0 on a faceted sphere) moves with the normals and is a jump that neither this pass nor the
    # reference's differentiates: -7 % at 16 x 32 facets, -4 % at 64 x 128
    assert got * fd > 0 and abs(got - fd) < 0.12 * abs(fd), (got, fd)


def test_interface_and_refusals():
    sc = build("emitter_in_view", 0.0, 16, 8)
    integ = epsm.load_dict({"type": "prb_reparam", "max_depth": 2})
    assert isinstance(integ, epsm.integrators.PRBReparamIntegrator) and integ.reparam is True
    assert (integ.reparam_rays, integ.reparam_kappa, integ.reparam_exp, integ.reparam_max_depth) == (16, 1e5, 3.0, 2)   # prb_reparam.py:233-250
    g = torch.ones((16, 16, 3))
    p = sc.param_grads()
    integ.render_backward(sc, p, g, sensor=0, seed=1, spp=8)            # nothing attached: nothing happens
    assert float(p.flat.abs().max()) == 0.0
    sc.attach("light", positions=True)
    p = sc.param_grads()
    integ.render_backward(sc, p, g, sensor=0, seed=1, spp=8)
    once = p.pos.clone()
    integ.render_backward(sc, p, g, sensor=0, seed=1, spp=8)            # gradients accumulate
    assert float(once.abs().max()) > 0 and torch.allclose(p.pos, 2 * once, rtol=1e-5, atol=1e-6)
    off = epsm.load_dict({"type": "prb_reparam", "max_depth": 2, "reparam_max_depth": 0})
    p0 = sc.param_grads()
    off.render_backward(sc, p0, g, sensor=0, seed=1, spp=8)             # no warp: an emitter on black has no other gradient
    assert float(p0.pos.abs().max()) < 1e-6 * float(once.abs().max())
    assert epsm.load_dict({"type": "prb_reparam", "reparam_antithetic": True}).reparam_antithetic is True
    assert integ.reparam_antithetic is False                                             # prb_reparam.py:243-246
    with pytest.raises(ValueError):
        epsm.load_dict({"type": "prb_reparam", "reparam_rays": 65})
    # a box reconstruction filter cannot carry image-space motion (common.py:379-388)
    d = {"type": "scene", "cam": sensor([0, 0, 4], [0, 0, 0], res=8, spp=4, rfilter="box"),
         "light": {"type": "mesh", "vertices": rect(0.5)[0], "faces": rect(0.5)[1], "face_normals": True,
                   "emitter": {"type": "area", "radiance": {"type": "rgb", "value": [1.0, 1.0, 1.0]}}}}
    sb = on_host(S.Scene.from_dict(d, device="cpu")); sb.tracer = "mega"
    sb.attach("light", positions=True)
    with pytest.raises(Exception, match="box reconstruction filter"):
        integ.render_backward(sb, sb.param_grads(), torch.ones((8, 8, 3)), sensor=0, seed=0, spp=4)


def test_sample_border_widens_the_wavefront_not_the_image():
    """film.sample_border (common.py:309-336): (width + 2 b)(height + 2 b) spp samples, the first of them b pixels outside."""
    sc = build("emitter_in_view", 0.0, 16, 4)
    s = sc.sensors[0]
    assert s.border == 2 and s.wavefront_size(4) == 20 * 20 * 4
    tr = sc._trace(0, 0, 4, 2, 0, 0, s.wavefront_size(4))
    fp = tr.film_pos
    assert float(fp.min()) >= -2.0 and float(fp.max()) <= 18.0 and float(fp[:4].max()) < -1.0
    assert tuple(sc.render_primal(sensor=0, seed=0, spp=4, max_depth=2).shape) == (16, 16, 3)
    plain = S.Scene.from_dict({"type": "scene", "cam": sensor([0, 0, 4], [0, 0, 0], res=16, spp=4, rfilter="gaussian")}, device="cpu")
    assert plain.sensors[0].border == 0
    # the manifold integrators' sensors have no border (epsm.py:239-246 maps path -> pixel without one)
    with pytest.raises(ValueError, match="sample border"):
        next(iter(sc.iter_traces(0, 0, 4, 3)))


def test_film_adjoint_against_autograd_of_the_splat():
    """integrators.film_adjoint_reparam = the adjoint of `image[p] = sum_i w_ip L_i det_i / sum_i w_ip det_i` (gaussian of
    stddev 0.5, radius 2, minus its value at the radius: src/rfilters/gaussian.cpp; common.py:888-903) w.r.t. every sample's
    radiance, film position and determinant: against torch autograd of that expression written densely."""
    from epsm_mitsuba3_amd.integrators import film_adjoint_reparam
    torch.manual_seed(3)
    H = W = 9
    n = 300
    pos = (torch.rand(n, 2, dtype=torch.float64) * (W + 3) - 1.5).requires_grad_(True)      # some samples in the border, some outside
    L = (torch.rand(n, 3, dtype=torch.float64) * 2).requires_grad_(True)
    det = torch.ones(n, dtype=torch.float64, requires_grad=True)
    grad_img = torch.randn(H, W, 3, dtype=torch.float64)
    radius, alpha = 2.0, -1.0 / (2.0 * 0.5 * 0.5)
    bias = math.exp(alpha * radius * radius)
    cx = torch.arange(W, dtype=torch.float64) + 0.5
    cy = torch.arange(H, dtype=torch.float64) + 0.5

    def w1(c, p):
        d = c[None, :] - p[:, None]
        return torch.where(d.abs() <= radius, (torch.exp(alpha * d * d) - bias).clamp_min(0), torch.zeros_like(d))
    wx, wy = w1(cx, pos[:, 0]), w1(cy, pos[:, 1])                      # (n,W), (n,H)
    w = wy[:, :, None] * wx[:, None, :] * det[:, None, None]          # (n,H,W)
    num = (w[..., None] * L[:, None, None, :]).sum(0)
    den = w.sum(0)
    image = num / den.clamp_min(1e-30)[..., None]
    (image * grad_img).sum().backward()
    accum = torch.cat([num, den[..., None]], -1).detach().float()
    dL, adj = film_adjoint_reparam(pos.detach().float(), L.detach().float(), grad_img.float(), accum)
    assert torch.allclose(dL.double(), L.grad, rtol=2e-4, atol=2e-5 * float(L.grad.abs().max()))
    want = torch.cat([pos.grad, det.grad[:, None]], 1)
    assert torch.allclose(adj.double(), want, rtol=2e-3, atol=2e-4 * float(want.abs().max())), float((adj.double() - want).abs().max())


def test_warp_adjoint_is_the_transpose_of_its_forward_mode():
    """<g, J u> = <J^T g, u>: the forward mode of one warp (origin moving with velocity u: V_theta, div V_theta,
    reparam.py:155-221) against the hand-written adjoint (reparam.py:269-333 in closed form, warp_backward) on the SAME
    auxiliary rays, and the same identity for a rigid translation of the mesh the rays hit (every hit point moves by u, the
    origin stays: the adjoint's vertex rows summed)."""
    v, f = rect(1.5, (0, 0, 2.0))
    d = {"type": "scene", "cam": sensor([0, 0, 4], [0, 0, 0], res=8),
         "wall": {"type": "mesh", "vertices": v, "faces": f[:, ::-1], "face_normals": True,
                  "bsdf": {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.5, 0.5, 0.5]}}}}
    sc = on_host(S.Scene.from_dict(d, device="cpu"))
    sc.attach("wall", positions=True)
    lib = host_tracer()
    rng = np.random.default_rng(0)
    o = np.array([0.2, -0.1, 0.0], np.float32)
    for trial in range(6):
        w = np.array([rng.normal() * 0.3, rng.normal() * 0.3, 1.0], np.float32); w /= np.linalg.norm(w)
        if trial >= 4:                                          # towards the rectangle's edge: some auxiliary rays miss
            w = np.array([1.5 - 0.2, 0.3, 2.0], np.float32) - o; w[0] += 0.003 * (trial - 4); w /= np.linalg.norm(w)
        u = rng.normal(size=3).astype(np.float32)
        g_dir = rng.normal(size=3).astype(np.float32); g_dir -= w * (w @ g_dir)
        g_div = np.float32(rng.normal())
        fwd = (C.c_float * 5)()
        adj = (C.c_float * 6)()
        gp = np.zeros((sc.V, 3), np.float32)
        ptr = lambda a: a.ctypes.data_as(C.c_void_p)
        kappa = 1e3 if trial >= 4 else 1e4
        lib.epsm_debug_warp(C.byref(sc.c_scene), ptr(o), ptr(w), ptr(u), 32, C.c_float(kappa), C.c_float(3.0), C.c_uint32(0), C.c_uint32(trial), fwd)
        lib.epsm_debug_warp_adjoint(C.byref(sc.c_scene), ptr(o), ptr(w), ptr(g_dir), C.c_float(g_div), 32, C.c_float(kappa), C.c_float(3.0),
                                    C.c_uint32(0), C.c_uint32(trial), ptr(gp), adj)
        lhs = float(g_dir @ np.array(fwd[:3]) + g_div * fwd[3])
        rhs = float(np.array(adj[:3]) @ u)
        assert abs(lhs - rhs) <= 2e-3 * max(abs(lhs), abs(rhs), 1e-3), (trial, lhs, rhs)
        # the mesh translated by u with the origin fixed moves every hit point as the origin moved by -u
        assert abs(float(gp.sum(0) @ u) + rhs) <= 2e-3 * max(abs(rhs), 1e-3), (trial, float(gp.sum(0) @ u), rhs)


def test_sensor_translation_gradient_on_the_host():
    """`prb_reparam` w.r.t. the sensor's position (TranslateCameraConfig, test_ad_integrators.py:639-674): every shape moves
    the other way as the sensor sees it, so the gradient is minus the sum of all vertex gradients (integrators.py).  The
    plumbing and the sign on the host build at a small sample count; the reference's thresholds at its sample count on the
    GPU (tests/test_gpu_reparam.py)."""
    got, fd, _ = fd_check("translate_camera_lit", device="cpu", spp=96, rays=16, seeds=1, fd_spp_mult=2)
    r, g, f = rel(got, fd)
    assert g * f > 0 and r < 0.6, (g, f)


def test_sensor_gradient_refuses_point_emitters():
    import epsm_mitsuba3_amd as epsm
    sc = build("receiver_point_light", 0.0, 16, 4, "cpu")
    sc.attach_sensor()
    integ = epsm.load_dict({"type": "prb_reparam", "max_depth": 2, "reparam_rays": 4})
    with pytest.raises(NotImplementedError):
        integ.render_backward(sc, sc.param_grads(), torch.ones((16, 16, 3)), sensor=0, seed=0, spp=4)


def test_film_crop_window_selects_the_rays_of_the_full_film():
    """hdrfilm crop_width / crop_height / crop_offset_* (CropWindowConfig, test_ad_integrators.py:250-281; sensor.h:227-262):
    the ray through the centre of pixel (i, j) of the window is the ray through pixel (i + ox, j + oy) of the full film, and
    its differentials are one WINDOW pixel wide."""
    import ctypes as C
    from _scenes import host_tracer
    from epsm_mitsuba3_amd import scene as S
    full = S.Sensor(sensor([0, 0, 4], [0, 0, 0], res=64, rfilter="gaussian"))
    d = sensor([0, 0, 4], [0, 0, 0], res=64, rfilter="gaussian")
    d["film"].update(crop_width=32, crop_height=24, crop_offset_x=32, crop_offset_y=20)
    crop = S.Sensor(d)
    assert (crop.width, crop.height) == (32, 24) and crop.wavefront_size(2) == 32 * 24 * 2
    lib = host_tracer()
    lib.epsm_probe.restype = C.c_int

    def ray(s_, x, y):
        rows = np.zeros((1, 8), np.float32); rows[0, :2] = (x, y)
        out = np.zeros((1, 16), np.float32)
        cs = s_.c_struct()
        assert lib.epsm_probe(C.c_int(8), C.c_int64(1), rows.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.byref(cs), None) == 0
        return out[0, :12].astype(np.float64)
    for i, j in ((0, 0), (5, 7), (31, 23), (16.5, 3.25)):
        a, b = ray(crop, i + 0.5, j + 0.5), ray(full, i + 32 + 0.5, j + 20 + 0.5)
        assert np.allclose(a, b, atol=2e-6), (i, j, a - b)
    with pytest.raises(ValueError):
        d["film"]["crop_offset_x"] = 40
        S.Sensor(d)
