"""GPU twin of tests/test_color_adjoint.py (the reference's FD recipe, test_ad_integrators.py:833-871, at its 32 x 32
image size) and the outer loop on colour parameters: ``prb`` alone and the ``*_hybrid`` switch of EPSM/optim.py:87-119."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rfilter,max_depth", [("gaussian", 4), ("box", 6)])
def test_color_adjoint_matches_finite_differences(rfilter, max_depth):
    from test_color_adjoint import fd_check, make_scene
    res, spp = 32, 128
    sc = make_scene(torch.device("cuda", 0), res, spp, rfilter)
    got, want, rel = fd_check(sc, res, spp, max_depth)
    assert float(want.abs().min()) > 0
    assert float(rel.mean()) < 0.05 and float(rel.max()) < 0.5, (got, want)      # the reference's thresholds
    assert float(rel.max()) < 0.03, (got, want)


def test_reflectances_are_recovered_by_the_colour_phase():
    from epsm_mitsuba3_amd.optim import run
    hist, opt = run("prb", "albedo", iterations=60, lr=0.03, log=lambda s: None)
    assert hist[0] > 0.6 and min(hist[-10:]) < 0.25 * hist[0], hist


def test_hybrid_switches_integrators_after_thres_iterations():
    """manifold_hybrid on the colour experiment: three manifold iterations (5-channel image, matcher; nothing of this
    scene's parameters is geometric, so nothing moves), then the optimiser is reset and ``prb_reparam`` takes over
    (3-channel image, L2 loss, colour adjoint) and the reflectances move towards the target."""
    from epsm_mitsuba3_amd.optim import run
    lines = []
    hist, opt = run("manifold_hybrid", "albedo", iterations=40, lr=0.03, log=lines.append)
    assert "phase 2 = PRBReparamIntegrator" in lines[0]
    assert abs(hist[3] - hist[0]) < 1e-6                      # the manifold phase has no colour gradient
    assert min(hist[-8:]) < 0.5 * hist[0], hist
