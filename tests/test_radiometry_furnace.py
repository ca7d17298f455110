"""Radiometric known answers that do not depend on any renderer: the furnace.  Inside a closed box whose walls all emit L
and reflect rho (diffuse), the radiance along every ray is L (1 + rho + ... + rho^(d-1)) for an integrator that follows
d - 1 bounces (common.py / epsm.py: ``max_depth = d``, at most six iterations, epsm.py:549) -- whatever mixture of emitter
sampling, BSDF sampling, MIS weights and Russian roulette produced it.  A lossless object inside changes nothing: a perfect
mirror returns the wall's radiance, and so does a glass body, whose two refractions scale the radiance by eta^-2 and eta^2
(dielectric.cpp:250-330).  Run on the host build of the per-path code here and on the GPU in test_gpu_radiometry.py."""
import numpy as np
import pytest
import torch

from _scenes import furnace, furnace_with_ball, on_host

RHO = np.array([0.5, 0.3, 0.8])


def check_diffuse_furnace(make, tracer, spp, pixel_tol, device="cpu"):
    # (max_depth > 6 stops inside the seventh term: its emitter-sampled part is in, its BSDF-sampled part is not -- epsm.py:549)
    for depth in (1, 2, 3, 5, 6):
        sc = make(furnace(reflectance=tuple(RHO), res=8, spp=spp, device=device))
        sc.tracer = tracer
        img = sc.render_primal(sensor=0, seed=3 + depth, spp=spp, max_depth=depth).cpu().double()
        expect = torch.tensor(sum(RHO ** i for i in range(depth)))
        rel = (img - expect) / expect
        assert float(rel.reshape(-1, 3).mean(0).abs().max()) < 0.01, (tracer, depth, img.reshape(-1, 3).mean(0), expect)
        assert float(rel.abs().max()) < pixel_tol, (tracer, depth)


def check_lossless_ball(make, tracer, spp, device="cpu"):
    # black walls: the only radiance in the box is the walls' own L = 1, and every ray must carry exactly that
    mirror = make(furnace_with_ball({"type": "conductor", "material": "none"}, spp=spp, device=device))
    mirror.tracer = tracer
    img = mirror.render_primal(sensor=0, seed=5, spp=spp, max_depth=4).cpu()
    assert float((img - 1.0).abs().max()) < 1e-4, tracer
    glass = make(furnace_with_ball({"type": "dielectric", "int_ior": 1.5, "ext_ior": 1.0}, spp=spp, device=device))
    glass.tracer = tracer
    img = glass.render_primal(sensor=0, seed=6, spp=spp, max_depth=6).cpu()
    # the middle of the ball: reflection, or refraction in and out, inside six iterations; towards the silhouette a facetted
    # ball traps light by total internal reflection for longer than the integrator follows it
    assert float((img[3:5, 3:5] - 1.0).abs().max()) < 2e-3, (tracer, img[3:5, 3:5, 0])
    assert float(img.max()) < 1.0 + 2e-3 and float(img.mean()) > 0.9
    # grey walls: the mirror ball shows the furnace value of the walls
    mirror = make(furnace_with_ball({"type": "conductor", "material": "none"}, wall_reflectance=0.3, spp=spp, device=device))
    mirror.tracer = tracer
    img = mirror.render_primal(sensor=0, seed=7, spp=spp, max_depth=6).cpu()
    expect = sum(0.3 ** i for i in range(6))
    assert abs(float(img.mean()) / expect - 1.0) < 0.01, (tracer, float(img.mean()), expect)


@pytest.mark.parametrize("tracer", ["mega", "wavefront"])
def test_diffuse_furnace_on_the_host_build(tracer):
    check_diffuse_furnace(on_host, tracer, spp=64, pixel_tol=0.08)


@pytest.mark.parametrize("tracer", ["mega", "wavefront"])
def test_lossless_ball_in_the_furnace_on_the_host_build(tracer):
    check_lossless_ball(on_host, tracer, spp=128)
