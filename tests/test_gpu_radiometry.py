"""tests/test_radiometry_furnace.py on the GPU, both tracer forms (the primal image of rows a1 / f1: emission, MIS,
BSDF sampling, Fresnel and the eta^2 scaling, path depth -- against closed forms, not against another build of the code)."""
import pytest
import torch

from test_radiometry_furnace import check_diffuse_furnace, check_lossless_ball

pytestmark = pytest.mark.gpu


def _as_is(scene):
    return scene


@pytest.mark.parametrize("tracer", ["mega", "wavefront"])
def test_diffuse_furnace(tracer):
    check_diffuse_furnace(_as_is, tracer, spp=256, pixel_tol=0.04, device="cuda")


@pytest.mark.parametrize("tracer", ["mega", "wavefront"])
def test_lossless_ball_in_the_furnace(tracer):
    check_lossless_ball(_as_is, tracer, spp=256, device="cuda")


@pytest.mark.parametrize("tracer", ["mega", "wavefront"])
def test_direct_illumination_of_a_diffuse_floor(tracer):
    """E = L A cos cos / d^2 under a small light, outgoing radiance rho / pi E: emitter sampling and BSDF sampling that
    hits the light, combined by MIS (tests/test_tracer_host.py has the same on the host build)."""
    import math
    from _scenes import floor_and_light
    L, half, h, rho = 50.0, 0.05, 2.0, 0.6
    sc = floor_and_light(L, half, h, rho, res=8, device="cuda")
    sc.tracer = tracer
    img = sc.render_primal(sensor=0, seed=2, spp=2048, max_depth=2).cpu()
    tr = sc._trace(0, seed=2, spp=64, max_depth=2, K=1, lo=0, hi=8 * 8 * 64)
    p = tr.path_info[1]["points"][3].cpu().numpy().reshape(8, 8, 64, 3).mean(axis=2)
    A = (2 * half) ** 2
    for (y, x) in ((3, 3), (3, 4), (4, 3), (4, 4)):
        q = p[y, x]
        d2 = q[0] ** 2 + q[1] ** 2 + h ** 2
        cos = h / math.sqrt(d2)
        expect = rho / math.pi * L * A * cos * cos / d2
        assert abs(float(img[y, x, 0]) - expect) / expect < 0.03, (y, x, float(img[y, x, 0]), expect)
