"""Analytic known-answer tests of the tracer's per-path code (epsm_trace_core.h), run on the host
build of tests/host_harness.  They pin what cannot be compared with Mitsuba here:
camera rays (perspective.cpp:238-279), the sampler (PCG32), direct illumination with emitter
sampling + MIS (epsm.py:569-605), delta reflection / refraction, and the vertex log (epsm.py:648-654)."""
import math

import numpy as np
import pytest
import torch

from _scenes import floor_and_light, on_host, quad, sensor
from epsm_mitsuba3_amd import scene as S
from epsm_mitsuba3_amd.records import loose_from_packed


def test_camera_rays_match_the_perspective_model():
    res, fov = 32, 50.0
    d = {"type": "scene", "cam": sensor([1.0, 2.0, 3.0], [1.0, 2.0, -5.0], up=(0, 1, 0), fov=fov, res=res, spp=1, near=0.5)}
    sc = on_host(S.Scene.from_dict(d, device="cpu"))
    tr = sc._trace(0, seed=3, spp=1, max_depth=1, K=0, lo=0, hi=res * res)
    o, dirs, dx, dy, pos = tr.ray_o.numpy(), tr.ray_d.numpy(), tr.ray_dx.numpy(), tr.ray_dy.numpy(), tr.film_pos.numpy()
    assert np.allclose(np.linalg.norm(dirs, axis=1), 1, atol=1e-5)
    # film position -> direction: x_cam = (0.5 - sx) * 2 tan(fov/2), looking down -z world (camera +z)
    t = math.tan(math.radians(fov / 2))
    sx, sy = pos[:, 0] / res, pos[:, 1] / res
    cam_dir = np.stack([(1 - 2 * sx) * t, (1 - 2 * sy) * t, np.ones_like(sx)], -1)
    cam_dir /= np.linalg.norm(cam_dir, axis=1, keepdims=True)
    W = S.look_at([1.0, 2.0, 3.0], [1.0, 2.0, -5.0], (0, 1, 0))[:3, :3]
    assert np.allclose(dirs, cam_dir @ W.T, atol=2e-5)
    # origin sits on the near plane: o = eye + d * near / d_z(cam)
    assert np.allclose(o, np.array([1.0, 2.0, 3.0]) + dirs * (0.5 / cam_dir[:, 2:3]), atol=1e-4)
    # ray differentials: one pixel to the right / down
    cam_dx = np.stack([(1 - 2 * (sx + 1 / res)) * t, (1 - 2 * sy) * t, np.ones_like(sx)], -1)
    cam_dx /= np.linalg.norm(cam_dx, axis=1, keepdims=True)
    assert np.allclose(dx, cam_dx @ W.T, atol=2e-5)
    cam_dy = np.stack([(1 - 2 * sx) * t, (1 - 2 * (sy + 1 / res)) * t, np.ones_like(sx)], -1)
    cam_dy /= np.linalg.norm(cam_dy, axis=1, keepdims=True)
    assert np.allclose(dy, cam_dy @ W.T, atol=2e-5)
    # pixel-major, row-major order with jitter inside the pixel (common.py:320-335)
    pix = np.arange(res * res)
    assert np.all(np.floor(pos[:, 0]) == pix % res) and np.all(np.floor(pos[:, 1]) == pix // res)


def test_sampler_is_deterministic_and_decorrelated():
    sc = on_host(S.Scene.from_dict({"type": "scene", "cam": sensor([0, 0, 3], [0, 0, 0], res=8, spp=8)}, device="cpu"))
    a = sc._trace(0, seed=5, spp=8, max_depth=1, K=0, lo=0, hi=512).film_pos
    b = sc._trace(0, seed=5, spp=8, max_depth=1, K=0, lo=0, hi=512).film_pos
    c = sc._trace(0, seed=6, spp=8, max_depth=1, K=0, lo=0, hi=512).film_pos
    # tiles reproduce the same paths (path_offset seeds by wavefront index)
    d = sc._trace(0, seed=5, spp=8, max_depth=1, K=0, lo=128, hi=256).film_pos
    assert torch.equal(a, b) and not torch.equal(a, c) and torch.equal(a[128:256], d)
    jitter = (a - torch.floor(a)).numpy()
    assert abs(jitter.mean() - 0.5) < 0.03 and abs(np.corrcoef(jitter[:, 0], jitter[:, 1])[0, 1]) < 0.1


def test_directly_visible_emitter_has_its_radiance():
    lv, lf = quad(0.0, 5.0, up=True)
    d = {"type": "scene", "cam": sensor([0, 0, 3], [0, 0, 0], res=8, spp=2),
         "light": {"type": "mesh", "vertices": lv, "faces": lf, "face_normals": True,
                   "emitter": {"type": "area", "radiance": {"type": "rgb", "value": [2.0, 3.0, 4.0]}}}}
    sc = on_host(S.Scene.from_dict(d, device="cpu"))
    img = sc.render_primal(sensor=0, seed=1, spp=2, max_depth=3)
    assert torch.allclose(img, torch.tensor([2.0, 3.0, 4.0]).expand_as(img), atol=1e-5)


def test_direct_illumination_of_a_diffuse_floor():
    """E = L A cos cos / d^2 for a small light; outgoing radiance rho/pi * E.  The two estimators
    (emitter sampling, BSDF sampling that hits the light) are combined by MIS and must add up."""
    L, half, h, rho = 50.0, 0.05, 2.0, 0.6
    sc = on_host(floor_and_light(L, half, h, rho, res=8))
    img = sc.render_primal(sensor=0, seed=2, spp=512, max_depth=2)
    # pixel (4,4) looks at ~the origin; take the central 2x2 pixels and their world hit points
    tr = sc._trace(0, seed=2, spp=64, max_depth=2, K=1, lo=0, hi=8 * 8 * 64)
    p = tr.path_info[1]["points"][3].numpy().reshape(8, 8, 64, 3).mean(axis=2)
    A = (2 * half) ** 2
    for (y, x) in ((3, 3), (3, 4), (4, 3), (4, 4)):
        q = p[y, x]
        d2 = q[0] ** 2 + q[1] ** 2 + h ** 2
        cos = h / math.sqrt(d2)
        expect = rho / math.pi * L * A * cos * cos / d2
        got = float(img[y, x, 0])
        assert abs(got - expect) / expect < 0.05, (y, x, got, expect)


def test_mirror_shows_the_light():
    """conductor with material 'none' is a perfect mirror (eta=0,k=1): a camera ray reflected into an
    area light carries its radiance; the emitter hit after a delta bounce gets MIS weight 1."""
    mv, mf = quad(0.0, 4.0, up=True)
    lv, lf = quad(3.0, 6.0, up=False)
    d = {"type": "scene", "cam": sensor([0, -1.0, 1.0], [0, 0, 0], up=(0, 0, 1), res=8, spp=4),
         "mirror": {"type": "mesh", "vertices": mv, "faces": mf, "face_normals": True,
                    "bsdf": {"type": "conductor", "material": "none"}},
         "light": {"type": "mesh", "vertices": lv, "faces": lf, "face_normals": True,
                   "emitter": {"type": "area", "radiance": {"type": "rgb", "value": 7.0}}}}
    sc = on_host(S.Scene.from_dict(d, device="cpu"))
    img = sc.render_primal(sensor=0, seed=0, spp=4, max_depth=3)
    assert torch.allclose(img[2:6, 2:6], torch.full((4, 4, 3), 7.0), rtol=1e-4)
    tr = sc._trace(0, seed=0, spp=4, max_depth=3, K=2, lo=0, hi=8 * 8 * 4)
    v1, v2 = tr.path_info[1], tr.path_info[2]
    act = (v1["active"] > 0) & (v2["active"] > 0)
    # law of reflection at the logged vertices: normalize(wi + wo) is the surface normal (0,0,1)
    cam = tr.path_info[0]["cam"]
    wi = torch.nn.functional.normalize(cam - v1["points"][3], dim=1)
    wo = torch.nn.functional.normalize(v2["points"][3] - v1["points"][3], dim=1)
    h = torch.nn.functional.normalize(wi + wo, dim=1)
    assert torch.allclose(h[act], torch.tensor([0.0, 0.0, 1.0]).expand(int(act.sum()), 3), atol=1e-4)
    assert bool((v1["bsdf"][act] & 0x20).all()) and bool((v1["eta"][act] == 1).all())


def test_glass_slab_obeys_snell_and_logs_eta():
    top_v, top_f = quad(1.0, 4.0, up=True)         # normal +z: outside above
    bot_v, bot_f = quad(0.0, 4.0, up=False)        # normal -z: outside below
    flo_v, flo_f = quad(-1.0, 6.0, up=True)
    glass = {"type": "dielectric", "int_ior": 1.5, "ext_ior": 1.0}
    d = {"type": "scene", "cam": sensor([0, -2.0, 3.0], [0, 0, 1.0], up=(0, 0, 1), res=8, spp=16),
         "top": {"type": "mesh", "vertices": top_v, "faces": top_f, "face_normals": True, "bsdf": glass},
         "bottom": {"type": "mesh", "vertices": bot_v, "faces": bot_f, "face_normals": True, "bsdf": glass},
         "floor": {"type": "mesh", "vertices": flo_v, "faces": flo_f, "face_normals": True, "bsdf": {"type": "diffuse"}}}
    sc = on_host(S.Scene.from_dict(d, device="cpu"))
    tr = sc._trace(0, seed=4, spp=16, max_depth=6, K=3, lo=0, hi=8 * 8 * 16)
    v1, v2, v3 = tr.path_info[1], tr.path_info[2], tr.path_info[3]
    cam = tr.path_info[0]["cam"]
    # paths that were transmitted at the top face, hit the bottom face next and left the slab
    entered = (v1["active"] > 0) & (v2["active"] > 0) & (v1["eta"] > 1.2) & (v2["points"][3][:, 2].abs() < 1e-4)
    assert int(entered.sum()) > 100
    din = torch.nn.functional.normalize(v1["points"][3] - cam, dim=1)[entered]
    dmid = torch.nn.functional.normalize(v2["points"][3] - v1["points"][3], dim=1)[entered]
    sin_i, sin_t = din[:, :2].norm(dim=1), dmid[:, :2].norm(dim=1)
    assert torch.allclose(sin_i, 1.5 * sin_t, rtol=2e-3, atol=1e-4)                  # Snell
    assert torch.allclose(v1["eta"][entered], torch.tensor(1.5), atol=1e-5)          # eta_it logged (epsm.py:653)
    left = entered & (v3["active"] > 0) & ((v2["eta"] - 1 / 1.5).abs() < 1e-4)
    assert int(left.sum()) > 50
    dout = torch.nn.functional.normalize(v3["points"][3] - v2["points"][3], dim=1)[left]
    din_l = torch.nn.functional.normalize(v1["points"][3] - cam, dim=1)[left]
    assert torch.allclose(dout, din_l, atol=2e-3)                                    # parallel slab: direction restored
    assert bool(((v1["bsdf"][entered] & 0x60) == 0x60).all())                         # DeltaReflection | DeltaTransmission


def test_vertex_log_is_consistent():
    sc = on_host(floor_and_light(res=8, bsdf={"type": "roughconductor", "material": "Al", "distribution": "ggx", "alpha": 0.2}))
    sc.attach("floor", positions=True)
    sc.attach_alpha("floor.bsdf")
    tr = sc._trace(0, seed=9, spp=32, max_depth=4, K=3, lo=0, hi=8 * 8 * 32)
    v1 = tr.path_info[1]
    act = v1["active"] > 0
    assert int(act.sum()) > 1000
    b0, b1 = v1["uv"]
    p = v1["points"][0] * b0[:, None] + v1["points"][1] * b1[:, None] + v1["points"][2] * (1 - b0 - b1)[:, None]
    assert torch.allclose(p[act], v1["points"][3][act], atol=1e-5)                    # epsm.py:758-759 reproduces si.p
    assert bool((v1["ismesh"][act] == 1).all()) and bool(((v1["bsdf"][act] & 0x8) != 0).all())
    # hf is the sampled microfacet normal: reflect(wi, m) = wo in the local frame <=> m || wi + wo
    hf = v1["hf"][act]
    assert torch.allclose(hf.norm(dim=1), torch.ones(int(act.sum())), atol=1e-4) and bool((hf[:, 2] > 0).all())
    # emitter sample point lies on the light (z = 2, |x|,|y| <= 0.05) whenever it is usable
    em = v1["active_em"] > 0
    lp = v1["light"][em]
    assert torch.allclose(lp[:, 2], torch.full((int(em.sum()),), 2.0), atol=1e-5) and bool((lp[:, :2].abs() <= 0.0501).all())
    # parameter addressing: triangle rows inside the floor mesh's slice, mode bits = attached flat mesh
    # (the log carries triangle ids; the scene's table turns them into vertex rows + mode bits, include/epsm.h)
    tlo, thi = sc.mesh_tri_slices["floor"]
    ids = tr.scatter_info[0]["tri"][act].long()
    assert bool(((ids >= tlo) & (ids < thi)).all())
    loose = loose_from_packed(tr.scatter_info[0])
    lo, hi = sc.mesh_slices["floor"]
    vidx = loose["vidx"][act]
    assert bool(((vidx >= lo) & (vidx < hi)).all()) and bool((loose["mode"][act] == 4).all())
    assert bool((loose["bsdf_id"][act] == 0).all())
    evidx = loose["evidx"][em]
    llo, lhi = sc.mesh_slices["light"]
    assert bool(((evidx >= llo) & (evidx < lhi)).all())
    # second vertex of an inactive path is logged as zeros / inactive
    v2 = tr.path_info[2]
    dead = v2["active"] == 0
    assert bool((v2["points"][0][dead] == 0).all()) and bool((v2["ismesh"][dead] == 0).all())


def test_rectangle_is_not_a_mesh():
    d = {"type": "scene", "cam": sensor([0, 0, 3], [0, 0, 0], fov=25, res=4, spp=2),
         "floor": {"type": "rectangle", "bsdf": {"type": "diffuse"}}}
    sc = on_host(S.Scene.from_dict(d, device="cpu"))
    tr = sc._trace(0, seed=0, spp=2, max_depth=2, K=1, lo=0, hi=32)
    v1 = tr.path_info[1]
    assert bool((v1["active"] == 1).all()) and bool((v1["ismesh"] == 0).all()) and bool((v1["points"][0] == 0).all())


def test_records_feed_the_gradient_oracle():
    """End of the chain on the CPU: traced records -> oracle tangent-free calc_grad gives finite,
    partly non-zero gradients for a specular floor under an area light."""
    from oracle.binding import oracle_calc_grad
    sc = on_host(floor_and_light(res=8, bsdf={"type": "roughconductor", "material": "Al", "distribution": "ggx", "alpha": 0.05}))
    tr = sc._trace(0, seed=1, spp=16, max_depth=3, K=2, lo=0, hi=8 * 8 * 16)
    N = tr.ray_d.shape[0]
    dlduv = torch.zeros((N, 1, 6)); dlduv[:, 0, :2] = 1e-4
    fp, lg, dg, _ = oracle_calc_grad("manifold", tr.path_info, dlduv, torch.zeros((N, 3)), dtype=torch.float64)
    g = torch.stack(fp)
    assert bool(torch.isfinite(g).all()) and float(g.abs().max()) > 0


def test_occluder_record_of_the_first_vertex():
    """epsm.py:609-620 (integrators with max_depth <= 3): the closest hit of the ray from the first vertex
    towards its emitter sample is logged with its barycentrics and dis = |ds.p - hit| / |ds.p - si.p|."""
    fv, ff = quad(0.0, 3.0, up=True)
    ov, of = quad(1.0, 0.4, up=True)                           # occluder plate at z = 1 above the origin
    lv, lf = quad(3.0, 0.05, up=False)                         # small light at z = 3
    d = {"type": "scene", "cam": sensor([0.0, -2.5, 2.0], [0, 0, 0], up=(0, 0, 1), res=16, spp=16),
         "floor": {"type": "mesh", "vertices": fv, "faces": ff, "face_normals": True, "bsdf": {"type": "diffuse"}},
         "plate": {"type": "mesh", "vertices": ov, "faces": of, "face_normals": True,
                   "bsdf": {"type": "twosided", "bsdf": {"type": "diffuse"}}},
         "light": {"type": "mesh", "vertices": lv, "faces": lf, "face_normals": True,
                   "emitter": {"type": "area", "radiance": {"type": "rgb", "value": 30.0}}}}
    sc = on_host(S.Scene.from_dict(d, device="cpu"))
    sc.attach("plate", positions=True)
    n = 16 * 16 * 16
    assert sc._trace(0, seed=2, spp=16, max_depth=4, K=2, lo=0, hi=n).scatter_info[0]["shadow"] is None   # max_depth > 3
    tr = sc._trace(0, seed=2, spp=16, max_depth=3, K=2, lo=0, hi=n)
    sh = tr.scatter_info[0]["shadow"]
    assert sh is not None and tr.scatter_info[1].get("shadow") is None
    v1 = tr.path_info[1]
    x = v1["points"][3]
    on_floor = (v1["active"] > 0) & (x[:, 2].abs() < 1e-5) & (v1["active_em"] > 0)
    assert tuple(sh.shape) == (n, 4)
    loose = loose_from_packed(tr.scatter_info[0])
    tri, fl = loose["svidx"], torch.stack([loose["sb0"], loose["sb1"], loose["sdis"]], dim=1)
    plo, phi = sc.mesh_slices["plate"]
    llo, lhi = sc.mesh_slices["light"]
    hit_plate = on_floor & ((tri >= plo) & (tri < phi)).all(1)
    hit_light = on_floor & ((tri >= llo) & (tri < lhi)).all(1)
    assert int(hit_plate.sum()) > 50 and int(hit_light.sum()) > 50
    # the ray towards the light ends on one of the two (samples on the light's very edge may slip past it)
    assert float((hit_plate | hit_light)[on_floor].double().mean()) > 0.98
    # unoccluded: the hit is the emitter sample itself, dis ~ 0 -> 0 (:615)
    assert bool((fl[hit_light, 2] == 0).all())
    # occluded: the hit lies on the plate (z = 1) on the segment floor point -> light sample, dis = 2/3 here
    verts = sc.vertex_positions("plate").double()
    c0, c1 = fl[hit_plate, 0].double(), fl[hit_plate, 1].double()
    t = tri[hit_plate] - plo
    p = verts[t[:, 0]] * c0[:, None] + verts[t[:, 1]] * c1[:, None] + verts[t[:, 2]] * (1 - c0 - c1)[:, None]
    lp, xp = v1["light"][hit_plate].double(), x[hit_plate].double()
    assert torch.allclose(p[:, 2], torch.ones_like(p[:, 2]), atol=1e-5)
    s = (p - xp).norm(dim=1) / (lp - xp).norm(dim=1)
    assert torch.allclose(xp + (lp - xp) * s[:, None], p, atol=1e-4)
    assert torch.allclose(fl[hit_plate, 2].double(), (lp - p).norm(dim=1) / (lp - xp).norm(dim=1), atol=1e-5)
    assert torch.allclose(fl[hit_plate, 2], torch.full((int(hit_plate.sum()),), 2.0 / 3.0), atol=0.02)
    assert bool((loose["smode"][hit_plate] == 4).all())                # the plate is a flat mesh with attached positions
    # paths without a usable emitter sample carry no occluder
    dead = ~((v1["active"] > 0) & (v1["active_em"] > 0))
    assert bool((sh[dead, 0] == -1).all()) and bool((fl[dead, 2] == 0).all())


def test_vertex_update_refits_the_bvh_and_recomputes_normals():
    """params.update(): `set_vertex_positions` overwrites the rows of the flat buffers on the device, recomputes
    the vertex normals and refits the BVH (same topology); tracing then sees exactly what a scene built from
    the moved vertices sees."""
    from epsm_mitsuba3_amd.exp.human_tube import SkinnedTube
    m = SkinnedTube("cpu", rings=12, sectors=10)
    fv, ff = quad(0.0, 4.0, up=True)
    lv, lf = quad(5.0, 0.3, up=False)

    def scene(verts):
        d = {"type": "scene", "cam": sensor([0.5, -4.0, 2.5], [0, 0, 1.0], up=(0, 0, 1), fov=45, res=24, spp=4),
             "floor": {"type": "mesh", "vertices": fv, "faces": ff, "face_normals": True, "bsdf": {"type": "diffuse"}},
             "tube": {"type": "mesh", "vertices": verts, "faces": m.faces, "bsdf": {"type": "diffuse"}},      # vertex normals
             "light": {"type": "mesh", "vertices": lv, "faces": lf, "face_normals": True,
                       "emitter": {"type": "area", "radiance": {"type": "rgb", "value": 20.0}}}}
        return on_host(S.Scene.from_dict(d, device="cpu"))

    v0 = m.gen_mesh(torch.zeros(1, 6))[0]
    v1 = m.gen_mesh(torch.tensor([[0.0, 0.5, 0.0, 0.3, -0.4, 0.0]]))[0] + torch.tensor([0.3, 0.2, 0.0])
    moved = scene(v0.numpy().astype(np.float64))
    nodes_before = moved.bvh.nodes.clone()
    moved.set_vertex_positions("tube", v1)
    assert not torch.equal(moved.bvh.nodes[:, :24], nodes_before[:, :24])                  # boxes changed,
    assert torch.equal(moved.bvh.nodes[:, 24:].view(torch.int32), nodes_before[:, 24:].view(torch.int32))   # topology kept
    fresh = scene(v1.numpy().astype(np.float64))
    lo, hi = fresh.mesh_slices["tube"]
    assert torch.allclose(moved.positions, fresh.positions, atol=1e-6)
    assert torch.allclose(moved.normals[lo:hi], fresh.normals[lo:hi], atol=1e-4)
    n = 24 * 24 * 4
    a = moved._trace(0, seed=3, spp=4, max_depth=3, K=2, lo=0, hi=n)
    b = fresh._trace(0, seed=3, spp=4, max_depth=3, K=2, lo=0, hi=n)
    v_first = loose_from_packed(a.scatter_info[0])["vidx"][:, 0]
    on_tube = (a.path_info[1]["active"] > 0) & (v_first >= lo) & (v_first < hi)
    assert int(on_tube.sum()) > 100
    assert torch.equal(a.path_info[1]["active"], b.path_info[1]["active"])
    assert torch.allclose(a.path_info[1]["points"][3], b.path_info[1]["points"][3], atol=1e-5)
    assert torch.allclose(a.path_info[1]["normal"], b.path_info[1]["normal"], atol=1e-4)
    assert torch.equal(a.scatter_info[0]["tri"], b.scatter_info[0]["tri"])
    assert torch.allclose(a.radiance, b.radiance, rtol=1e-4, atol=1e-5)
    # the host copy of the mesh follows when it is needed again (attach -> full upload)
    moved.attach("tube", positions=True)
    assert np.allclose(moved.mesh("tube").v, v1.numpy(), atol=1e-6)
