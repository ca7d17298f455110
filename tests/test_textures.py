"""`bitmap` reflectance textures of the native tracer (src/textures/bitmap.cpp; EPSM/exp/human.py:117-124, glassslab.py:188-195) on the
host build: a diffuse rectangle under a uniform environment shows rho(uv) L, with rho looked up as bitmap.cpp:366-418 does
(uv * res - 0.5, bilinear or nearest, indices wrapped) -- restated in numpy here; texture coordinates of an inline mesh and of an
.obj file (v flipped, obj.cpp:267)."""
import math
import os

import numpy as np
import pytest
import torch

from _scenes import on_host, sensor
from epsm_mitsuba3_amd import scene as S
from test_environment import camera_dirs


def texture(H=16, W=24):
    j, i = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    a = np.stack([0.2 + 0.6 * i / (W - 1), 0.2 + 0.6 * j / (H - 1), 0.5 + 0.3 * np.sin(0.15 * i + 0.1 * j)], -1)      # (smooth: a pixel covers ~1.7 texels here)
    return a.astype(np.float32)


def lookup(tex, u, v, nearest):
    H, W = tex.shape[:2]
    x, y = u * W - 0.5, v * H - 0.5
    if nearest:
        return tex[np.floor(y + 0.5).astype(int) % H, np.floor(x + 0.5).astype(int) % W]
    i0, j0 = np.floor(x).astype(int), np.floor(y).astype(int)
    fx, fy = (x - i0)[..., None], (y - j0)[..., None]
    g = lambda jj, ii: tex[jj % H, ii % W].astype(np.float64)
    return g(j0, i0) * (1 - fx) * (1 - fy) + g(j0, i0 + 1) * fx * (1 - fy) + g(j0 + 1, i0) * (1 - fx) * fy + g(j0 + 1, i0 + 1) * fx * fy


def check_textured_plane(make, tracer, nearest, device="cpu", spp=128):
    tex = texture()
    L = np.array([1.0, 0.5, 2.0])
    # a rectangle [-1,1]^2 at z = 0 whose uv runs over [0,2] x [0,1.5]: the texture repeats
    v = np.array([[-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0]], float)
    uv = np.array([[0, 0], [2, 0], [2, 1.5], [0, 1.5]], float)
    origin, res, fov = [0.2, -0.1, 3.0], 32, 35
    d = {"type": "scene", "cam": sensor(origin, [0, 0, 0], fov=fov, res=res, spp=spp),
         "plane": {"type": "mesh", "vertices": v, "faces": np.array([[0, 1, 2], [0, 2, 3]]), "texcoords": uv, "face_normals": True,
                   "bsdf": {"type": "diffuse", "reflectance": {"type": "bitmap", "bitmap": tex,
                                                               "filter_type": "nearest" if nearest else "bilinear"}}},
         "sky": {"type": "constant", "radiance": {"type": "rgb", "value": list(L)}}}
    sc = make(S.Scene.from_dict(d, device=device))
    sc.tracer = tracer
    img = sc.render_primal(sensor=0, seed=2, spp=spp, max_depth=2).cpu().double().numpy()
    dirs = camera_dirs(origin, [0, 0, 0], (0, 1, 0), fov, res)
    o = np.asarray(origin)
    hit = o + dirs * (-o[2] / dirs[..., 2])[..., None]
    on = (np.abs(hit[..., 0]) < 0.9) & (np.abs(hit[..., 1]) < 0.9)
    u, w = (hit[..., 0] + 1), (hit[..., 1] + 1) * 0.75
    want = lookup(tex, u, w, nearest) * L
    assert on.sum() > 300
    H, W = tex.shape[:2]
    # not across the seam of the repeat (the texture jumps there and a pixel averages over its footprint) ...
    tx, ty = (u * W - 0.5) % W, (w * H - 0.5) % H                        # the seam lies between texel W - 1 and texel 0
    on &= (tx > 1.0) & (tx < W - 3.0) & (ty > 1.0) & (ty < H - 3.0)
    if nearest:          # ... nor, with the nearest filter, across a texel's edge
        fx, fy = (u * W) % 1.0, (w * H) % 1.0
        on &= (np.abs(fx - 0.5) < 0.3) & (np.abs(fy - 0.5) < 0.3)
    assert on.sum() > 150
    err = np.abs(img[on] / want[on] - 1)
    # (the Monte-Carlo noise of the emitter / BSDF samples at this sample count: ~2 % per pixel)
    assert float(np.median(err)) < 0.02 and err.max() < 0.12, (tracer, nearest, float(np.median(err)), float(err.max()))
    assert abs(float((img[on] / want[on]).mean()) - 1) < 0.005


@pytest.mark.parametrize("tracer", ["mega", "wavefront"])
@pytest.mark.parametrize("nearest", [False, True])
def test_textured_plane_shows_the_texture(tracer, nearest):
    check_textured_plane(on_host, tracer, nearest)


def test_obj_texture_coordinates_and_refusals(tmp_path):
    p = tmp_path / "quad.obj"
    p.write_text("\n".join(["v -1 -1 0", "v 1 -1 0", "v 1 1 0", "v -1 1 0", "vt 0 0", "vt 1 0", "vt 1 1", "vt 0 1", "f 1/1 2/2 3/3", "f 1/1 3/3 4/4", ""]))
    v, n, f, uv = S.load_obj(str(p), with_uv=True)
    assert v.shape == (4, 3) and n is None and np.allclose(uv, [[0, 1], [1, 1], [1, 0], [0, 0]])      # v flipped: obj.cpp:267
    sc = S.Scene.from_dict({"type": "scene", "cam": sensor([0, 0, 3], [0, 0, 0], res=8),
                            "q": {"type": "obj", "filename": str(p), "face_normals": True,
                                  "bsdf": {"type": "diffuse", "reflectance": {"type": "bitmap", "bitmap": texture()}}}}, device="cpu")
    assert sc.texcoords is not None and tuple(sc.texcoords.shape) == (4, 2) and sc.c_scene.n_textures == 1
    assert sc.meshes[0].flags() & S.MESH_HAS_UV
    with pytest.raises(ValueError):
        sc.attach_color("q.bsdf")                                         # a texture is not a colour parameter
    with pytest.raises(ValueError):
        S.Scene.from_dict({"type": "scene", "q": {"type": "rectangle", "bsdf": {"type": "diffuse", "reflectance": {"type": "bitmap", "filename": "wood2.jpg"}}}}, device="cpu")
