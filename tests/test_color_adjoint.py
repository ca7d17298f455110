"""The colour adjoint of the hybrid scheme's second phase (integrators.PRBIntegrator, epsm_trace_paths_color) by the
reference's own recipe for AD integrators (src/integrators/tests/test_ad_integrators.py:833-871): backward-mode
gradients of a small scene against finite differences of the rendered image under the same seed, thresholds mean 5 % /
max 50 % there -- here per parameter.  Runs the host build of the tracer (tests/host_harness); the GPU twin is
tests/test_gpu_color_adjoint.py."""
import numpy as np
import pytest
import torch

from _scenes import on_host, quad, sensor
from epsm_mitsuba3_amd import scene as S
import epsm_mitsuba3_amd as epsm


def make_scene(device="cpu", res=16, spp=64, rfilter="gaussian", floor=(0.6, 0.4, 0.3), wall=(0.2, 0.5, 0.7), light=(30.0, 25.0, 20.0)):
    fv, ff = quad(0.0, 2.0, up=True)
    wv = np.array([[-2, 2, 0], [2, 2, 0], [2, 2, 2.5], [-2, 2, 2.5]], float)          # a wall behind: interreflection
    wf = np.array([[0, 2, 1], [0, 3, 2]])
    lv, lf = quad(2.2, 0.4, up=False)
    d = {"type": "scene", "cam": sensor([0.0, -3.5, 1.6], [0, 0.5, 0.5], up=(0, 0, 1), res=res, spp=spp, rfilter=rfilter),
         "floor": {"type": "mesh", "vertices": fv, "faces": ff, "face_normals": True,
                   "bsdf": {"type": "diffuse", "reflectance": {"type": "rgb", "value": list(floor)}}},
         "wall": {"type": "mesh", "vertices": wv, "faces": wf, "face_normals": True,
                  "bsdf": {"type": "twosided", "bsdf": {"type": "diffuse", "reflectance": {"type": "rgb", "value": list(wall)}}}},
         "light": {"type": "mesh", "vertices": lv, "faces": lf, "face_normals": True,
                   "emitter": {"type": "area", "radiance": {"type": "rgb", "value": list(light)}}}}
    sc = S.Scene.from_dict(d, device=device)
    if str(device) == "cpu":
        on_host(sc)
    sc.tracer = "mega"
    return sc


def make_constant_scene(device="cpu", res=16, spp=64, albedo=(0.6, 0.4, 0.3), sky=(1.0, 0.8, 1.5)):
    """DiffuseAlbedoConfig / DiffuseAlbedoGIConfig / ConstantEmitterRadianceConfig of the reference's list
    (test_ad_integrators.py:116-160, 232-247): diffuse geometry under a `constant` environment emitter -- the derivative w.r.t.
    the albedo and w.r.t. the emitter's radiance (slot 1, a scale of it: `set_color` keeps the chromaticity test simple)."""
    fv, ff = quad(0.0, 1.5, up=True)
    wv = np.array([[-1.5, 1.5, 0], [1.5, 1.5, 0], [1.5, 1.5, 2.0], [-1.5, 1.5, 2.0]], float)
    wf = np.array([[0, 2, 1], [0, 3, 2]])
    bsdf = {"type": "twosided", "bsdf": {"type": "diffuse", "reflectance": {"type": "rgb", "value": list(albedo)}}}
    d = {"type": "scene", "cam": sensor([0.0, -3.5, 1.6], [0, 0.5, 0.5], up=(0, 0, 1), res=res, spp=spp, rfilter="gaussian"),
         "floor": {"type": "mesh", "vertices": fv, "faces": ff, "face_normals": True, "bsdf": bsdf},
         "wall": {"type": "mesh", "vertices": wv, "faces": wf, "face_normals": True, "bsdf": bsdf},
         "sky": {"type": "constant", "radiance": {"type": "rgb", "value": list(sky)}}}
    sc = S.Scene.from_dict(d, device=device)
    if str(device) == "cpu":
        on_host(sc)
    sc.tracer = "mega"
    return sc


def fd_check(sc, res, spp, max_depth, seed=3, h=2e-3, attach=None):
    integ = epsm.load_dict({"type": "prb_reparam", "max_depth": max_depth})
    assert isinstance(integ, epsm.integrators.PRBIntegrator) and integ.reparam is True      # the colour adjoint is its base class
    if attach is not None:
        slots = attach(sc)
        n_slots = len(slots)
    else:
        slots = [sc.attach_color("floor.bsdf"), sc.attach_color("wall.bsdf"), sc.attach_radiance("light")]
        assert slots == [0, 1, 2]
        n_slots = 3
    g = torch.Generator().manual_seed(1)
    grad_in = (0.5 + torch.rand((res, res, 3), generator=g)).to(sc.device)
    img = integ.render(sc, sensor=0, seed=seed, spp=spp)
    assert tuple(img.shape) == (res, res, 3) and float(img.max()) > 0
    params = sc.param_grads()
    assert tuple(params.color.shape) == (n_slots, 3)
    integ.render_backward(sc, params, grad_in, sensor=0, seed=seed, spp=spp)
    got = params.color.clone().cpu()
    vals = sc.color_values().cpu()
    want = torch.zeros_like(got)
    for j in range(n_slots):
        for c in range(3):
            out = []
            for sgn in (+1, -1):
                v = vals[j].clone(); v[c] *= (1 + sgn * h)
                sc.set_color(j, v.tolist())
                out.append(float((integ.render(sc, sensor=0, seed=seed, spp=spp) * grad_in).sum()))
            sc.set_color(j, vals[j].tolist())
            want[j, c] = (out[0] - out[1]) / (2 * h * float(vals[j, c]))
    rel = (got - want).abs() / want.abs().clamp_min(1e-6)
    return got, want, rel


@pytest.mark.parametrize("rfilter,max_depth", [("gaussian", 4), ("box", 3)])
def test_color_adjoint_matches_finite_differences_on_the_host_tracer(rfilter, max_depth):
    res, spp = 12, 48
    sc = make_scene("cpu", res, spp, rfilter)
    got, want, rel = fd_check(sc, res, spp, max_depth)
    assert float(want.abs().min()) > 0                       # every parameter matters in this scene
    assert float(rel.mean()) < 0.05 and float(rel.max()) < 0.5, (got, want)      # the reference's thresholds
    assert float(rel.max()) < 0.02, (got, want)              # radiance is multiplicative in these parameters: FD is exact up to fp32


@pytest.mark.parametrize("max_depth", [2, 3])
def test_albedo_and_radiance_under_a_constant_environment(max_depth):
    res, spp = 12, 48
    sc = make_constant_scene("cpu", res, spp)
    env = next(i for i, e in enumerate(sc.emitter_desc) if e["type"] == 2)
    got, want, rel = fd_check(sc, res, spp, max_depth, attach=lambda s_: [s_.attach_color("floor.bsdf"), s_.attach_radiance(env)])
    assert float(want.abs().min()) > 0
    assert float(rel.max()) < 0.02, (got, want)


def test_accumulation_and_registry():
    sc = make_scene("cpu", 8, 8, "box")
    sc.attach_radiance("light")
    integ = epsm.load_dict({"type": "prb", "max_depth": 3})
    p = sc.param_grads()
    g = torch.ones((8, 8, 3))
    integ.render_backward(sc, p, g, sensor=0, seed=1, spp=8)
    once = p.color.clone()
    integ.render_backward(sc, p, g, sensor=0, seed=1, spp=8)
    assert torch.allclose(p.color, 2 * once) and float(once.abs().max()) > 0
    # d sum(image) / d radiance * radiance = sum(image): the image is linear in the only light's radiance
    img = integ.render(sc, sensor=0, seed=1, spp=8)
    assert torch.allclose((once[0] * sc.color_values()[0]).cpu(), img.sum(dim=(0, 1)).cpu(), rtol=1e-3)


def test_crop_window_config():
    """CropWindowConfig (test_ad_integrators.py:250-281): a diffuse rectangle under the constant emitter seen through a 32 x 32
    window at offset (32, 20) of a 64 x 64 film with a sample border; the parameter is the plane's reflectance.  The image has
    the window's size, the film splat and its adjoint work on the window, and the derivative matches finite differences at the
    reference's thresholds (mean 5 %, max 50 %; the radiance is linear in the albedo at max_depth 2, so FD is exact here)."""
    v = np.array([[-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0]], float)
    f = np.array([[0, 1, 2], [0, 2, 3]])
    cam = sensor([0, 0, 4], [0, 0, 0], up=(0, 1, 0), fov=28.8415, res=64, spp=16, rfilter="gaussian", sample_border=True)
    cam["film"].update(crop_width=32, crop_height=32, crop_offset_x=32, crop_offset_y=20)
    d = {"type": "scene", "cam": cam,
         "plane": {"type": "mesh", "vertices": v, "faces": f, "face_normals": True,
                   "bsdf": {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.5, 0.4, 0.6]}}},
         "light": {"type": "constant"}}
    sc = on_host(S.Scene.from_dict(d, device="cpu"))
    sc.tracer = "mega"
    assert sc.sensors[0].wavefront_size(16) == (32 + 4) * (32 + 4) * 16            # the border surrounds the WINDOW
    got, want, rel = fd_check(sc, 32, 16, 2, attach=lambda s_: [s_.attach_color("plane.bsdf")])
    assert float(want.abs().min()) > 0
    assert float(rel.mean()) < 0.05 and float(rel.max()) < 0.5 and float(rel.max()) < 0.02, (got, want)
    # the window shows the right part of the plane: its left columns see the plane, the rightmost ones the background only
    integ = epsm.load_dict({"type": "prb", "max_depth": 2})
    img = integ.render(sc, sensor=0, seed=1, spp=16)
    full = dict(cam); full["film"] = {k: v_ for k, v_ in cam["film"].items() if not k.startswith("crop_")}
    d2 = dict(d); d2["cam"] = full
    sc2 = on_host(S.Scene.from_dict(d2, device="cpu")); sc2.tracer = "mega"
    img2 = integ.render(sc2, sensor=0, seed=1, spp=64)[20:52, 32:64]
    assert tuple(img.shape) == (32, 32, 3)
    assert float((img - img2).abs().mean()) < 0.06 * float(img2.abs().mean())     # the same picture up to sampling noise (16 vs 64 spp)
