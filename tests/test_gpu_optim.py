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


def test_plate_translation_is_recovered_under_an_environment_map():
    """The same with an `envmap` fill light (EPSM/exp/highlight.py:218-222): two emitters, half of the emitter samples go to the
    sky and log a far point without parameter rows."""
    from epsm_mitsuba3_amd.optim import run
    hist, opt = run("manifold", "highlight", iterations=45, lr=0.03, log=lambda s: None)
    assert hist[0] > 0.7
    assert min(hist[-10:]) < 0.3 * hist[0], hist


def test_three_coloured_lights_are_recovered():
    """EPSM/exp/glossyball.py: three lights of different colours over a glossy plate, an envmap as fill light; the matcher's colour
    channels tell the highlights apart, every light is pulled towards its own target."""
    from epsm_mitsuba3_amd.optim import run
    hist, opt = run("manifold", "glossyball", iterations=60, lr=0.03, log=lambda s: None)
    assert hist[0] > 0.55                                                # mean distance of the three lights from their targets: 0.60
    assert min(hist[-10:]) < 0.25 * hist[0], hist                        # measured: 0.073


def test_camera_translation_is_recovered_through_the_ray_origins():
    """The shape of EPSM/exp/bedroom.py (`trans2` of the camera): the only gradient is d / d ray.o = -sum grad_d (epsm.py:260-261),
    `ParamGrads.cam_origin` -- its sign and scale carry an Adam loop from 0.43 to 0.02 of distance."""
    from epsm_mitsuba3_amd.optim import run
    hist, opt = run("manifold", "camera", iterations=60, lr=0.02, log=lambda s: None)
    assert hist[0] > 0.4
    assert min(hist[-10:]) < 0.15 * hist[0] and hist[-1] < 0.25 * hist[0], hist


def test_caustic_light_translation_is_recovered():
    """manifold_caustic: camera -> diffuse floor -> glass slab (two refractions) -> area light; the light's
    gradient arrives through diffuse_grad of the chain's end point (epsm.py:1178-1184)."""
    from epsm_mitsuba3_amd.optim import run
    hist, opt = run("manifold_caustic", "slab", iterations=45, lr=0.03, log=lambda s: None)
    assert hist[0] > 0.55
    assert min(hist[-15:]) < 0.35 * hist[0], hist


def test_caustic_through_a_glass_sphere_is_recovered():
    """The stand-in for the reference's `manifold_caustic` box experiments (EPSM/all.sh:7,11: cornellbox, egg) and the shape of
    BASELINE.json configs[0]: camera -> diffuse floor -> two refractions on a smoothly shaded glass sphere -> area light.
    Measured: 0.43 -> 0.05-0.06."""
    from epsm_mitsuba3_amd.optim import run
    hist, opt = run("manifold_caustic", "caustic_sphere", iterations=60, lr=0.02, log=lambda s: None)
    assert hist[0] > 0.4
    assert min(hist[-15:]) < 0.3 * hist[0], hist


def test_shadow_occluder_translation_is_recovered():
    """max_depth = 2, everything diffuse: the only gradient path is the occluder term of epsm.py:609-620
    (first hit = floor point in or near the shadow, occluder = closest hit towards the emitter sample)."""
    from epsm_mitsuba3_amd.optim import run
    hist, opt = run("manifold", "shadow", iterations=45, lr=0.03, log=lambda s: None)
    assert hist[0] > 0.55
    assert min(hist[-15:]) < 0.35 * hist[0], hist


def test_tube_pose_is_recovered_through_per_vertex_gradients():
    """Round 1's form of config 5 (optim_human.py): vertices come from a torch module (three-bone skinned tube standing in for
    SMPL); the per-vertex gradients of the `human` mesh are chained into its pose with sum(verts * grad).backward()
    -- first-hit term on the figure itself plus the occluder term of its shadow (max_depth = 3)."""
    from epsm_mitsuba3_amd.optim import run
    hist, opt = run("manifold", "human_tube", iterations=80, lr=0.03, log=lambda s: None)
    assert hist[0] > 0.6
    assert min(hist[-15:]) < 0.35 * hist[0], hist


def test_body_pose_converges_and_stays_at_the_reference_settings():
    """Config 5 as the reference runs it (optim_human.py): 72 pose angles -> exp/body_model.py (SMPL's function, 6 890
    vertices, 7 829 in the atlas) -> renderer; backward sensor 256 x 256 @ 8 spp = 524 288 paths; per-vertex gradients
    chained with sum(verts * grad).backward(); Adam lr 0.01, match_Sinkhorn, pose clamped to +-0.1, primal and differential
    seeds de-correlated (util.py:505-513); the first 200 of its 1000 iterations.  The assertions are on the END of the history
    and on its minimum, at what is measured plus ~5 % (a threshold must not encode a loop that got 15 % worse):
    6.0 -> 2.5-2.9 cm around iteration 30 (the loop is chaotic in the order of its float atomics: 0.42-0.48 of the start over eight
    runs of three builds of the kernel, round 5), then 3.4-3.5 cm and staying (last-20 mean 0.58-0.60,
    maximum of the last 60 iterations 0.61).  Why it stays there is decided in profiles/r04_i_human_fd.txt: at the
    start EPSM's pose field is a descent direction of the matcher's own loss (cosine 0.77 with its finite differences), at the
    plateau it is not (-0.34 .. -0.45 on the best-determined angles the clamp does not hold) -- the occluder term, which the
    reference marks "not finished yet" (epsm.py:609), has grown as large as the first-hit term and opposes the loss's gradient:
    the fixed point is the field's own, not a stationary point of the loss."""
    import numpy as np
    from epsm_mitsuba3_amd.optim import run
    from epsm_mitsuba3_amd.exp import human
    assert (human.matcher, human.lr, human.it, human.thres) == ("Sinkhorn", 0.01, 1000, 1200)      # exp/human.py:6-11, optim_human.py:57,105
    hist, opt = run("manifold", "human", iterations=200, log=lambda s: None)                       # 200 of its 1000 iterations (0.25 s each)
    assert len(hist) == 201 and 0.05 < hist[0] < 0.07
    print("last-20 mean / start", np.mean(hist[-20:]) / hist[0], "max of last 60 / start", max(hist[-60:]) / hist[0], "best", min(hist) / hist[0])
    assert np.mean(hist[-20:]) < 0.63 * hist[0] and max(hist[-60:]) < 0.65 * hist[0], hist[::10]
    assert min(hist) < 0.51 * hist[0], hist[::10]
    assert float(opt["pose"].detach().abs().max()) <= human.POSE_CLAMP + human.lr * 1.5


def test_body_pose_is_recovered_by_the_hybrid_scheme():
    """`manifold_hybrid` with the switch inside the run: 40 manifold iterations, then prb_reparam's gradients of the L2
    image loss through the same skinning module (EPSM/optim_human.py:66-74, 100-114): ends below 20 % of the initial vertex
    distance and stays (measured 10 %)."""
    import numpy as np
    from epsm_mitsuba3_amd.optim import run
    from epsm_mitsuba3_amd.exp import human
    old = human.thres, human.spp
    human.thres, human.spp = 40, 16
    try:
        hist, opt = run("manifold_hybrid", "human", iterations=140, log=lambda s: None)
    finally:
        human.thres, human.spp = old
    print("at the switch", hist[40] / hist[0], "last-20 mean / start", np.mean(hist[-20:]) / hist[0])
    assert np.mean(hist[-20:]) < 0.2 * hist[0] and max(hist[-20:]) < 0.25 * hist[0], hist[::10]


def test_objects_seen_in_a_mirror_are_moved_onto_their_targets():
    """The reference's headline configuration (exp/bathroom.py): three tiles whose coloured side is visible only in
    a mirror are translated until the mirror image matches -- camera -> mirror (delta) -> diffuse object; the
    gradient arrives through diffuse_grad[1] of the continuing sub-path."""
    from epsm_mitsuba3_amd.optim import run
    hist, opt = run("manifold", "bathroom", iterations=60, lr=0.02, log=lambda s: None)
    assert hist[0] > 0.65
    assert min(hist[-10:]) < 0.25 * hist[0], hist
