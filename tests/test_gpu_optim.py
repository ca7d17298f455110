"""End to end: the outer loop (EPSM/optim.py shape) recovers the translation of an area light from the
position of its highlight on a specular plate (the `li` parameters of EPSM/exp/highlight.py) -- exercises
tracer, tangent, fused gradient/scatter (through the emitter rows, epsm.py:622-627), matcher and Adam
together and pins the SIGN conventions along the whole chain."""
import pytest

pytestmark = pytest.mark.gpu


def test_plate_translation_is_recovered():
    from epsm_mitsuba3_amd.optim import run
    hist, opt = run("manifold", "plate", iterations=45, lr=0.03, log=lambda s: None)
    assert hist[0] > 0.7
    assert min(hist[-10:]) < 0.25 * hist[0], hist


def test_caustic_light_translation_is_recovered():
    """manifold_caustic: camera -> diffuse floor -> glass slab (two refractions) -> area light; the light's
    gradient arrives through diffuse_grad of the chain's end point (epsm.py:1178-1184)."""
    from epsm_mitsuba3_amd.optim import run
    hist, opt = run("manifold_caustic", "slab", iterations=45, lr=0.03, log=lambda s: None)
    assert hist[0] > 0.55
    assert min(hist[-15:]) < 0.35 * hist[0], hist


def test_shadow_occluder_translation_is_recovered():
    """max_depth = 2, everything diffuse: the only gradient path is the occluder term of epsm.py:609-620
    (first hit = floor point in or near the shadow, occluder = closest hit towards the emitter sample)."""
    from epsm_mitsuba3_amd.optim import run
    hist, opt = run("manifold", "shadow", iterations=45, lr=0.03, log=lambda s: None)
    assert hist[0] > 0.55
    assert min(hist[-15:]) < 0.35 * hist[0], hist


def test_human_pose_is_recovered_through_per_vertex_gradients():
    """Config 5 (optim_human.py): vertices come from a torch module (three-bone skinned tube standing in for
    SMPL); the per-vertex gradients of the `human` mesh are chained into its pose with sum(verts * grad).backward()
    -- first-hit term on the figure itself plus the occluder term of its shadow (max_depth = 3)."""
    from epsm_mitsuba3_amd.optim import run
    hist, opt = run("manifold", "human", iterations=80, lr=0.03, log=lambda s: None)
    assert hist[0] > 0.6
    assert min(hist[-15:]) < 0.35 * hist[0], hist


def test_objects_seen_in_a_mirror_are_moved_onto_their_targets():
    """The reference's headline configuration (exp/bathroom.py): three tiles whose coloured side is visible only in
    a mirror are translated until the mirror image matches -- camera -> mirror (delta) -> diffuse object; the
    gradient arrives through diffuse_grad[1] of the continuing sub-path."""
    from epsm_mitsuba3_amd.optim import run
    hist, opt = run("manifold", "bathroom", iterations=60, lr=0.02, log=lambda s: None)
    assert hist[0] > 0.65
    assert min(hist[-10:]) < 0.25 * hist[0], hist
