"""Environment emitters on the GPU: the closed forms of tests/test_environment.py through the device build, both tracer forms,
and the backward pass on a scene lit by a map (the emitter sample of such a path is a far point without parameter rows:
epsm.py:622-627 finds nothing to follow)."""
import numpy as np
import pytest
import torch

import epsm_mitsuba3_amd as epsm
from _reparam_scenes import sphere
from _scenes import on_host, sensor
from epsm_mitsuba3_amd import scene as S
from test_environment import check_mirror_and_background, check_plane_under_a_map, check_uniform_environment, rot, smooth_map

pytestmark = pytest.mark.gpu
same = lambda sc: sc


@pytest.mark.parametrize("tracer", ["mega", "wavefront"])
def test_environment_closed_forms_on_the_device(tracer):
    check_uniform_environment(same, tracer, device="cuda")
    check_mirror_and_background(same, tracer, device="cuda")
    check_plane_under_a_map(same, tracer, device="cuda", spp=4096)


@pytest.mark.parametrize("tracer", ["mega", "wavefront"])
def test_bitmap_textures_on_the_device(tracer):
    from test_textures import check_textured_plane
    for nearest in (False, True):
        check_textured_plane(same, tracer, nearest, device="cuda", spp=256)


def scene_dict(res, spp):
    v, n, f = sphere(0.6, (0.1, 0.0, 0.0), 12, 24)
    fv = np.array([[-3, -1, -3], [3, -1, -3], [3, -1, 3], [-3, -1, 3]], float)
    return {"type": "scene", "cam": sensor([0, 0.5, 4], [0, 0, 0], fov=35, res=res, spp=spp),
            "ball": {"type": "mesh", "vertices": v, "normals": n, "faces": f, "bsdf": {"type": "roughconductor", "alpha": 0.15, "distribution": "ggx"}},
            "floor": {"type": "mesh", "vertices": fv, "faces": np.array([[0, 2, 1], [0, 3, 2]]), "face_normals": True,
                      "bsdf": {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.6, 0.6, 0.6]}}},
            "sky": {"type": "envmap", "bitmap": smooth_map(), "to_world": rot((0, 1, 0), 30.0)}}


def test_device_image_and_log_equal_the_host_build():
    res, spp = 16, 8
    host = on_host(S.Scene.from_dict(scene_dict(res, spp), device="cpu"))
    dev = S.Scene.from_dict(scene_dict(res, spp), device="cuda")
    for tracer in ("mega", "wavefront"):
        host.tracer = dev.tracer = tracer
        a = host.render_primal(sensor=0, seed=4, spp=spp, max_depth=4)
        b = dev.render_primal(sensor=0, seed=4, spp=spp, max_depth=4).cpu()
        # (a ray that grazes an edge may land on either side in the two builds: a handful of pixels differ by one sample)
        assert float(((a - b).abs() > 1e-3 * (1 + a.abs())).float().mean()) < 0.03, tracer
    ta = host._trace(0, 4, spp, 4, 3, 0, res * res * spp)
    tb = dev._trace(0, 4, spp, 4, 3, 0, res * res * spp)
    for k in (1, 2):
        la, lb = ta.path_info[k]["light"], tb.path_info[k]["light"].cpu()
        ok = ta.path_info[k]["active_em"].bool() & tb.path_info[k]["active_em"].cpu().bool()
        assert int(ok.sum()) > 100
        assert float(((la[ok] - lb[ok]).norm(dim=1) > 1e-3 * la[ok].norm(dim=1)).float().mean()) < 0.02, k


@pytest.mark.parametrize("name", ["manifold", "manifold_caustic"])
def test_backward_pass_under_a_map_is_finite_and_repeatable(name):
    res, spp = 32, 16
    sc = S.Scene.from_dict(scene_dict(res, spp), device="cuda")
    sc.attach("ball", positions=True, normals=True)
    sc.attach("floor", positions=True)
    integ = epsm.load_dict({"type": name, "max_depth": 4})
    g = torch.randn((res, res, 5), generator=torch.Generator().manual_seed(1)).cuda() * 1e-2
    out = []
    for _ in range(2):
        p = sc.param_grads()
        integ.render_backward(sc, p, g, sensor=0, seed=7, spp=spp)
        torch.cuda.synchronize()
        assert bool(torch.isfinite(p.flat).all())
        out.append(p.flat.clone())
    assert float(out[0].abs().max()) > 0
    assert float((out[0] - out[1]).abs().max()) <= 1e-5 * float(out[0].abs().max())
