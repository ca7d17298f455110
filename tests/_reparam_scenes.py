"""Restatements of the reference's test scenes for reparameterised integrators (src/integrators/tests/
test_ad_integrators.py:268-640) with the plugins this library has: obj meshes become inline meshes (a lat-long sphere with
vertex normals, a two-triangle rectangle with face normals).  The configs of the reference that are lit by a `constant`
environment emitter come twice: with an area light in its place (rounds 1-3, before the tracer had that emitter) and, as
`*_constant`, as the reference states them.  ``build(name, theta)`` translates the config's moving meshes by
``theta`` along x, as TranslateShapeConfigBase.update does."""
import numpy as np

from _scenes import on_host, sensor
from epsm_mitsuba3_amd import scene as S


def sphere(radius=1.0, center=(0, 0, 0), n_lat=16, n_lon=32):
    v, n, f = [], [], []
    for i in range(n_lat + 1):
        th = np.pi * i / n_lat
        for j in range(n_lon):
            ph = 2 * np.pi * j / n_lon
            d = np.array([np.sin(th) * np.cos(ph), np.cos(th), np.sin(th) * np.sin(ph)])
            v.append(d * radius + np.asarray(center, float)); n.append(d)
    for i in range(n_lat):
        for j in range(n_lon):
            a, b = i * n_lon + j, i * n_lon + (j + 1) % n_lon
            c, d_ = a + n_lon, b + n_lon
            if i > 0:
                f.append([a, b, c])
            if i < n_lat - 1:
                f.append([b, d_, c])
    return np.array(v), np.array(n), np.array(f)


def rect(half=1.0, center=(0, 0, 0), normal="+z"):
    v = np.array([[-half, -half, 0], [half, -half, 0], [half, half, 0], [-half, half, 0]], float)
    f = np.array([[0, 1, 2], [0, 2, 3]])
    if normal == "+x":
        v = v[:, [2, 0, 1]]
    return v + np.asarray(center, float), f


def plane_texture(H=32, W=32):
    """A texture with structure at the scale of the image: stripes and a blob over a gradient."""
    j, i = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    x, y = (i + 0.5) / W, (j + 0.5) / H
    a = np.stack([0.3 + 0.5 * x + 0.15 * np.sin(12 * x), 0.25 + 0.5 * y, 0.4 + 0.4 * np.exp(-((x - 0.6) ** 2 + (y - 0.4) ** 2) / 0.03)], -1)
    return a.astype(np.float32)


def gradient_map(H=24, W=48):
    """A smooth lat-long map with a strong dependence on direction (factor ~6 between its sides) and no symmetry."""
    j, i = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    th, ph = np.pi * j / (H - 1), 2 * np.pi * (i + 0.5) / W
    a = np.stack([1.0 + 0.8 * np.cos(ph) * np.sin(th), 0.8 + 0.6 * np.cos(th), 0.7 + 0.6 * np.sin(ph + 0.7) * np.sin(th)], -1)
    return a.astype(np.float32)


CONFIGS = {
    # TranslateRectangleEmitterOnBlackConfig (:383-410): an emitting rectangle, partly in view, on black
    "rectangle_emitter_on_black": dict(max_depth=2, moving=["light"], fd_eps=1e-3),
    # the same rectangle, smaller and wholly in view: with ramp weights 0.5 + x / width the answer is known in closed form
    # (the image of the rectangle shifts): pixels_per_unit * area_in_pixels / width
    "emitter_in_view": dict(max_depth=2, moving=["light"], fd_eps=1e-3),
    # TranslateSphereEmitterOnBlackConfig (:413-435)
    "sphere_emitter_on_black": dict(max_depth=2, moving=["light"], fd_eps=1e-3),
    # ScaleSphereEmitterOnBlackConfig (:438-460): vertex positions * (1 + theta)
    "scale_sphere_emitter_on_black": dict(max_depth=3, moving=["light"], fd_eps=1e-3, motion="scale"),
    # TranslateSelfShadowAreaLightConfig (:551-596): a plane and an upright rectangle on it move TOGETHER under a point
    # light and a dim constant environment; max_depth 3
    "self_shadow_point_light": dict(max_depth=3, moving=["plane", "occluder"], fd_eps=1e-3),
    # TranslateOccluderAreaLightConfig (:463-500): a small sphere between a small bright light and a diffuse plane
    "occluder_area_light": dict(max_depth=2, moving=["occluder"], fd_eps=2e-4, kappa=5e5),
    # a diffuse sphere in front of a lit diffuse wall, lit by an area light: silhouette + shading + shadow
    "diffuse_sphere_area_light": dict(max_depth=3, moving=["sphere"], fd_eps=1e-3),
    # TranslateSphereOnGlossyFloorConfig (:601-640) with an area light instead of the constant emitter
    "sphere_on_glossy_floor": dict(max_depth=3, moving=["sphere"], fd_eps=1e-3, kappa=2e5),
    # no discontinuity in view: a diffuse plane larger than the image moves ALONG ITS NORMAL under a small light -- the
    # change of the geometric term reaches the gradient only through the emitter ray's warp field and its divergence
    "receiver_along_normal": dict(max_depth=2, moving=["plane"], fd_eps=5e-3, dir=(0, 0, 1)),
    "receiver_point_light": dict(max_depth=2, moving=["plane"], fd_eps=5e-3, dir=(0, 0, 1)),
    # the same with interreflection between two walls (three vertices: the `extra` terms of the neighbours' BSDFs)
    "corner_along_normal": dict(max_depth=3, moving=["plane"], fd_eps=5e-3, dir=(0, 0, 1)),
    # TranslateDiffuseSphereConstantConfig (:317-338), TranslateDiffuseRectangleConstantConfig (:341-363): a diffuse body in a
    # uniform environment -- all the gradient there is comes from the silhouette against the background
    "diffuse_sphere_constant": dict(max_depth=2, moving=["sphere"], fd_eps=1e-3),
    "diffuse_rectangle_constant": dict(max_depth=2, moving=["rectangle"], fd_eps=8e-4),
    # TranslateShadowReceiverAreaLightConfig (:482-520) as committed there: the light is the constant emitter, the PLANE moves
    "shadow_receiver_constant": dict(max_depth=2, moving=["plane"], fd_eps=1e-3),
    # TranslateSphereOnGlossyFloorConfig (:600-637) as stated: constant emitter of radiance 1
    "sphere_on_glossy_floor_constant": dict(max_depth=3, moving=["sphere"], fd_eps=1e-3, kappa=2e5),
    # the first of them under an `envmap` whose radiance varies strongly with direction: the background behind the silhouette
    # and the light on the body both depend on where the reparameterised rays point
    # TranslateTexturedPlaneConfig (:523-550): a 2 x 2 rectangle (scale 2 of the unit one) with a `bitmap` reflectance under the
    # constant emitter, res 64 there; its museum.exr is not in the repository: a smooth synthetic texture stands in
    "textured_plane_constant": dict(max_depth=2, moving=["plane"], fd_eps=1e-3, res=64),
    # the same plane larger than the image: no silhouette in view, the image changes only because the texture slides with the plane
    "textured_plane_fills_the_view": dict(max_depth=2, moving=["plane"], fd_eps=2e-3),
    # TranslateCameraConfig (:639-674): a sphere under the constant emitter, the SENSOR moves along its own x axis
    # (to_world @ translate(theta, 0, 0)); res 16, spp 1024, max_depth 2, 64 rays, kappa 1e4 there
    "translate_camera": dict(max_depth=2, moving=[], camera=True, fd_eps=1e-3, res=16, kappa=1e4),
    # the same with an area light and a floor: shading, shadow and silhouettes all move in the image when the sensor does
    "translate_camera_lit": dict(max_depth=3, moving=[], camera=True, fd_eps=2e-3, res=32),
    "diffuse_sphere_envmap": dict(max_depth=2, moving=["sphere"], fd_eps=1e-3),
    "glossy_sphere_envmap": dict(max_depth=2, moving=["sphere"], fd_eps=1e-3),
}


def build(name, theta=0.0, res=32, spp=64, device="cpu", theta_n=0.0):
    """``theta``: translation of the moving meshes; ``theta_n``: their vertex normals become n + theta_n * NRM_DIR (not
    renormalised: the interpolation normalises, mesh.cpp:795-797)."""
    off = theta * np.asarray(CONFIGS[name].get("dir", (1.0, 0.0, 0.0)), float)
    if CONFIGS[name].get("motion") == "scale":
        off = 0.0
    cam = sensor([0, 0, 4], [0, 0, 0], up=(0, 1, 0), fov=28.8415, res=res, spp=spp, rfilter="gaussian", sample_border=True)   # mi default fov; film as in test_ad_integrators.py:60-70
    if CONFIGS[name].get("camera"):          # the sensor moves along ITS x axis: to_world @ translate(theta, 0, 0)  (:669-672)
        cam["to_world"] = np.asarray(cam["to_world"], float) @ S.translate([theta, 0.0, 0.0])
        off = 0.0
    d = {"type": "scene", "cam": cam}
    white = {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.5, 0.5, 0.5]}}
    if name == "rectangle_emitter_on_black":
        v, f = rect(1.0, (1.25, 0, 0))
        d["light"] = {"type": "mesh", "vertices": v + off, "faces": f, "face_normals": True,
                      "emitter": {"type": "area", "radiance": {"type": "rgb", "value": [1.0, 1.0, 1.0]}}}
    elif name == "emitter_in_view":
        v, f = rect(0.5, (0.1, -0.05, 0))
        d["light"] = {"type": "mesh", "vertices": v + off, "faces": f, "face_normals": True,
                      "emitter": {"type": "area", "radiance": {"type": "rgb", "value": [1.0, 1.0, 1.0]}}}
    elif name == "sphere_emitter_on_black":
        v, n, f = sphere(1.0, (1.25, 0, 0))
        d["light"] = {"type": "mesh", "vertices": v + off, "normals": n, "faces": f,
                      "emitter": {"type": "area", "radiance": {"type": "rgb", "value": [1.0, 1.0, 1.0]}}}
    elif name == "scale_sphere_emitter_on_black":
        v, n, f = sphere(1.0, (0, 0, 0))
        d["light"] = {"type": "mesh", "vertices": v * (1.0 + theta), "normals": n, "faces": f,
                      "emitter": {"type": "area", "radiance": {"type": "rgb", "value": [1.0, 1.0, 1.0]}}}
    elif name == "self_shadow_point_light":
        v, f = rect(1.0)
        d["plane"] = {"type": "mesh", "vertices": v + off, "faces": f, "face_normals": True, "bsdf": white}
        v, f = rect(1.0, (-1.0, 0, 0.5), "+x")
        v = (v - np.array([-1.0, 0, 0.5])) * np.array([1.0, 1.0, 0.5]) + np.array([-1.0, 0, 0.5])     # 2 x 1, standing on the plane
        d["occluder"] = {"type": "mesh", "vertices": v + off, "faces": f, "face_normals": True,
                         "bsdf": {"type": "twosided", "bsdf": white}}
        d["light"] = {"type": "point", "position": [-4.0, 0.0, 6.0], "intensity": {"type": "rgb", "value": [5.0, 0.0, 0.0]}}
        d["light2"] = {"type": "constant", "radiance": 0.1}
    elif name == "occluder_area_light":
        v, f = rect(1.0)
        d["plane"] = {"type": "mesh", "vertices": v, "faces": f, "face_normals": True, "bsdf": white}
        v, n, f = sphere(0.25, (2.0, 0, 2.0))
        d["occluder"] = {"type": "mesh", "vertices": v + off, "normals": n, "faces": f, "bsdf": white}
        v, n, f = sphere(0.05, (4.0, 0, 4.0), 8, 16)
        d["light"] = {"type": "mesh", "vertices": v, "normals": n, "faces": f,
                      "emitter": {"type": "area", "radiance": {"type": "rgb", "value": [1000.0, 1000.0, 1000.0]}}}
    elif name == "diffuse_sphere_area_light":
        v, f = rect(3.0, (0, 0, -1.0))
        d["wall"] = {"type": "mesh", "vertices": v, "faces": f, "face_normals": True, "bsdf": white}
        v, n, f = sphere(0.5, (0.2, 0.1, 0.3))
        d["sphere"] = {"type": "mesh", "vertices": v + off, "normals": n, "faces": f,
                       "bsdf": {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.9, 0.5, 0.1]}}}
        v, f = rect(0.7, (1.5, 2.0, 3.0))
        d["light"] = {"type": "mesh", "vertices": v, "faces": f[:, ::-1], "face_normals": True,
                      "emitter": {"type": "area", "radiance": {"type": "rgb", "value": [20.0, 20.0, 20.0]}}}
    elif name == "sphere_on_glossy_floor":
        v, f = rect(4.0)
        c, s = np.cos(np.radians(-45)), np.sin(np.radians(-45))
        R = np.array([[1, 0, 0], [0, c, -s], [0, s, c]])
        d["floor"] = {"type": "mesh", "vertices": (R @ v.T).T + np.array([0, 1.5, 0]) * 0 + np.array([0, -0.5, 0]), "faces": f, "face_normals": True,
                      "bsdf": {"type": "roughconductor", "alpha": 0.025}}
        v, n, f = sphere(0.5, (0.3, 0.6, 0.8))
        d["sphere"] = {"type": "mesh", "vertices": v + off, "normals": n, "faces": f,
                       "bsdf": {"type": "diffuse", "reflectance": {"type": "rgb", "value": [1.0, 0.5, 0.0]}}}
        v, f = rect(2.0, (0, 3.0, 4.0))
        d["light"] = {"type": "mesh", "vertices": v, "faces": f[:, ::-1], "face_normals": True,
                      "emitter": {"type": "area", "radiance": {"type": "rgb", "value": [5.0, 5.0, 5.0]}}}
    elif name in ("receiver_along_normal", "corner_along_normal", "receiver_point_light"):
        v, f = rect(3.0)
        d["plane"] = {"type": "mesh", "vertices": v + off, "faces": f, "face_normals": True, "bsdf": white}
        if name == "corner_along_normal":
            v, f = rect(3.0, (-1.2, 0, 0), "+x")
            d["side"] = {"type": "mesh", "vertices": v, "faces": f, "face_normals": True,
                         "bsdf": {"type": "twosided", "bsdf": {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.8, 0.3, 0.3]}}}}
        v, f = rect(0.2, (0.8, 0.3, 1.5))
        if name == "receiver_point_light":
            d["light"] = {"type": "point", "position": [0.8, 0.3, 1.5], "intensity": {"type": "rgb", "value": [6.0, 6.0, 6.0]}}
        else:
            d["light"] = {"type": "mesh", "vertices": v, "faces": f[:, ::-1], "face_normals": True,
                          "emitter": {"type": "area", "radiance": {"type": "rgb", "value": [40.0, 40.0, 40.0]}}}
    elif name == "diffuse_sphere_constant":
        v, n, f = sphere(1.0, (0, 0, 0))
        d["sphere"] = {"type": "mesh", "vertices": v + off, "normals": n, "faces": f, "bsdf": white}
        d["light"] = {"type": "constant"}
    elif name == "diffuse_rectangle_constant":
        v, f = rect(1.0)
        d["rectangle"] = {"type": "mesh", "vertices": v + off, "faces": f, "face_normals": True, "bsdf": white}
        d["light"] = {"type": "constant"}
    elif name == "shadow_receiver_constant":
        v, f = rect(1.0)
        d["plane"] = {"type": "mesh", "vertices": v + off, "faces": f, "face_normals": True, "bsdf": white}
        v, n, f = sphere(0.25, (2.0, 0, 2.0))
        d["occluder"] = {"type": "mesh", "vertices": v, "normals": n, "faces": f, "bsdf": white}
        d["light"] = {"type": "constant"}
    elif name == "sphere_on_glossy_floor_constant":
        v, f = rect(4.0)
        c, s = np.cos(np.radians(-45)), np.sin(np.radians(-45))
        R = np.array([[1, 0, 0], [0, c, -s], [0, s, c]])
        d["floor"] = {"type": "mesh", "vertices": (R @ v.T).T + np.array([0, 1.5, 0]), "faces": f, "face_normals": True,
                      "bsdf": {"type": "roughconductor", "alpha": 0.025}}
        v, n, f = sphere(1.0, (0.5, 2.0, 1.5))
        d["sphere"] = {"type": "mesh", "vertices": v + off, "normals": n, "faces": f,
                       "bsdf": {"type": "diffuse", "reflectance": {"type": "rgb", "value": [1.0, 0.5, 0.0]}}}
        d["light"] = {"type": "constant", "radiance": 1.0}
    elif name in ("textured_plane_constant", "textured_plane_fills_the_view"):
        v, f = rect(2.0 if name == "textured_plane_constant" else 4.0)
        uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], float) * (1.0 if name == "textured_plane_constant" else 2.0)
        d["plane"] = {"type": "mesh", "vertices": v + off, "faces": f, "texcoords": uv, "face_normals": True,
                      "bsdf": {"type": "diffuse", "reflectance": {"type": "bitmap", "bitmap": plane_texture()}}}
        d["light"] = {"type": "constant"}
    elif name == "translate_camera":
        v, n, f = sphere(1.0, (0, 0, 0))
        d["sphere"] = {"type": "mesh", "vertices": v, "normals": n, "faces": f, "bsdf": white}
        d["light"] = {"type": "constant"}
    elif name == "translate_camera_lit":
        v, f = rect(3.0, (0, 0, -0.6))
        d["floor"] = {"type": "mesh", "vertices": v, "faces": f, "face_normals": True, "bsdf": white}
        v, n, f = sphere(0.5, (0.3, 0.1, 0.2))
        d["sphere"] = {"type": "mesh", "vertices": v, "normals": n, "faces": f,
                       "bsdf": {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.9, 0.5, 0.1]}}}
        v, f = rect(0.5, (1.0, 1.5, 3.0))
        d["light"] = {"type": "mesh", "vertices": v, "faces": f[:, ::-1], "face_normals": True,
                      "emitter": {"type": "area", "radiance": {"type": "rgb", "value": [30.0, 30.0, 30.0]}}}
    elif name in ("diffuse_sphere_envmap", "glossy_sphere_envmap"):
        v, n, f = sphere(1.0, (0, 0, 0))
        bsdf = white if name.startswith("diffuse") else {"type": "roughconductor", "alpha": 0.3, "distribution": "ggx"}
        d["sphere"] = {"type": "mesh", "vertices": v + off, "normals": n, "faces": f, "bsdf": bsdf}
        d["light"] = {"type": "envmap", "bitmap": gradient_map(), "to_world": S.rotate([0.3, 1.0, 0.2], 40.0)}
    else:
        raise KeyError(name)
    if theta_n:
        for m in CONFIGS[name]["moving"]:
            d[m]["normals"] = np.asarray(d[m]["normals"], float) + theta_n * np.asarray(NRM_DIR)
    sc = S.Scene.from_dict(d, device=device)
    if str(device) == "cpu":
        on_host(sc)
    sc.tracer = "mega"
    return sc


NRM_DIR = (0.3, -0.2, 0.25)


def fd_check_normals(name, device="cpu", spp=128, rays=16, fd_eps=2e-2, fd_spp_mult=2):
    """d sum(image * ramp) / d theta_n through ``params.nrm`` against central differences."""
    import torch

    import epsm_mitsuba3_amd as epsm
    cfg = CONFIGS[name]
    res = cfg.get("res", 32)
    integ = epsm.load_dict({"type": "prb_reparam", "max_depth": cfg["max_depth"], "reparam_rays": rays})
    sc = build(name, 0.0, res, spp, device)
    for m in cfg["moving"]:
        sc.attach(m, positions=False, normals=True)
    g = torch.ones((res, res, 3), device=sc.device) * (0.5 + torch.arange(res, device=sc.device, dtype=torch.float32) / res)[None, :, None]
    params = sc.param_grads()
    integ.render_backward(sc, params, g, sensor=0, seed=0, spp=spp)
    u = torch.tensor(NRM_DIR, device=sc.device, dtype=torch.float32)
    got = 0.0
    for m in cfg["moving"]:            # the loader normalises what it is given: n(theta) = normalize(n + theta u), dn = (I - n n^T) u
        n = sc.vertex_normals(m)
        got += float((params.mesh_nrm(m) * (u[None, :] - n * (n @ u)[:, None])).sum())
    assert float(params.pos.abs().max()) == 0.0                      # positions are not attached
    v = []
    for sgn in (1, -1):
        s2 = build(name, 0.0, res, spp * fd_spp_mult, device, theta_n=sgn * fd_eps)
        v.append(float((integ.render(s2, sensor=0, seed=100, spp=spp * fd_spp_mult) * g).sum()))
    return got, (v[0] - v[1]) / (2 * fd_eps)


def fd_check(name, device="cpu", spp=128, seeds=1, rays=32, weights="ramp", fd_spp_mult=4, fd_eps=0.0, kappa=0.0, reparam_depth=-1):
    """The reference's recipe (test_ad_integrators.py:833-871): d sum(image * weights) / d theta by ``render_backward``
    against central differences of the primal image under common random numbers.  Returns (gradients per seed, finite
    differences per seed, seconds of the backward passes)."""
    import time

    import torch

    import epsm_mitsuba3_amd as epsm
    cfg = CONFIGS[name]
    res = cfg.get("res", 32)
    props = {"type": "prb_reparam", "max_depth": cfg["max_depth"], "reparam_rays": rays, "reparam_kappa": kappa or cfg.get("kappa", 1e5)}
    if reparam_depth >= 0:
        props["reparam_max_depth"] = reparam_depth
    integ = epsm.load_dict(props)
    sc = build(name, 0.0, res, spp, device)
    for m in cfg["moving"]:
        sc.attach(m, positions=True, normals=False)
    if cfg.get("camera"):
        sc.attach_sensor()
    g = torch.ones((res, res, 3), device=sc.device)
    if weights == "ramp":       # a shadow or an object that merely MOVES inside the image changes this loss at first order
        g = g * (0.5 + torch.arange(res, device=sc.device, dtype=torch.float32) / res)[None, :, None]
    u = torch.tensor(cfg.get("dir", (1.0, 0.0, 0.0)), device=sc.device, dtype=torch.float32)
    got, t0 = [], time.time()
    for seed in range(seeds):
        params = sc.param_grads()
        integ.render_backward(sc, params, g, sensor=0, seed=seed, spp=spp)
        if cfg.get("camera"):                 # d / d theta of to_world @ translate(theta, 0, 0): the sensor's x axis in the world
            ax = torch.tensor(np.asarray(sc.sensors[0].to_world, float)[:3, 0], device=sc.device, dtype=torch.float32)
            got.append(float((params.cam_origin * ax).sum()))
            assert float(params.pos.abs().max()) == 0.0          # no mesh was attached by the caller
        elif cfg.get("motion") == "scale":      # p(theta) = p (1 + theta)
            got.append(sum(float((params.mesh_pos(m) * sc.vertex_positions(m)).sum()) for m in cfg["moving"]))
        else:
            got.append(sum(float((params.mesh_pos(m) @ u).sum()) for m in cfg["moving"]))
    dt = time.time() - t0
    h = fd_eps or cfg.get("fd_eps", 1e-3)
    fd = []
    for seed in range(seeds):
        v = []
        for sgn in (1, -1):
            s2 = build(name, sgn * h, res, spp * fd_spp_mult, device)
            v.append(float((integ.render(s2, sensor=0, seed=100 + seed, spp=spp * fd_spp_mult) * g).sum()))
        fd.append((v[0] - v[1]) / (2 * h))
    return got, fd, dt
