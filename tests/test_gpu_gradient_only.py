"""EPSM_TRACE_GRADIENT_ONLY on the device (include/epsm_trace.h): `render_backward` with and without it gives the same
parameter gradients on the experiment scenes -- both tracer forms, both variants, the occluder term (max_depth <= 3)
included -- and the per-bounce queues of the wavefront tracer do shrink."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _scenes(dev):
    from epsm_mitsuba3_amd.exp import bathroom, clutter, human, plate, shadow, slab
    out = []
    sc = clutter.load_scene(dev, n_spheres=40, res=128, spp=8)
    for i in range(0, 40, 3):
        sc.attach(f"s{i}", positions=True, normals=True)
    sc.attach("floor", positions=True)
    out.append(("clutter", sc, clutter.max_depth))
    for mod in (plate, slab, shadow, bathroom, human):
        sc = mod.load_scene(dev)
        mod.optim_settings(sc)                         # attaches what the experiment optimises
        out.append((mod.__name__.rsplit(".", 1)[-1], sc, mod.max_depth))
    return out


@pytest.fixture(scope="module")
def scenes():
    assert torch.cuda.is_available()
    return _scenes(torch.device("cuda", 0))


@pytest.mark.parametrize("tracer", ["wavefront", "mega"])
@pytest.mark.parametrize("variant", ["manifold", "manifold_caustic"])
def test_render_backward_is_unchanged_by_the_flag(scenes, variant, tracer):
    import epsm_mitsuba3_amd as epsm
    dev = torch.device("cuda", 0)
    for name, sc, max_depth in scenes:
        sc.tracer = tracer
        res = sc.sensors[2].width
        g = torch.Generator().manual_seed(7)
        grad_in = (torch.randn((res, res, 5), generator=g) * 1e-3).to(dev)
        out = []
        for flag in (False, True):
            integ = epsm.load_dict({"type": variant, "max_depth": max_depth, "gradient_only": flag})
            p = sc.param_grads()
            integ.render_backward(sc, p, grad_in, seed=11)
            torch.cuda.synchronize()
            out.append(p.flat.double().cpu())
        m = float(out[0].abs().max())
        if variant == "manifold":
            assert m > 0, name
        # the same terms, summed by atomics in another order (fixed-point rows: differences come from the few float rows only)
        assert float((out[0] - out[1]).abs().max()) <= 1e-5 * m + 1e-12, (name, m, float((out[0] - out[1]).abs().max()))


def test_flag_needs_a_log_and_its_modifier_needs_the_flag(scenes):
    import ctypes as C
    from epsm_mitsuba3_amd import _lib, scene as S
    _, sc, _ = scenes[1]
    lib = _lib.lib()
    n = 64
    dev = sc.device
    ray = torch.empty((4, n, 3), device=dev)
    rad = torch.empty((n, 3), device=dev)
    fn = lib.epsm_trace_paths
    fn.restype = C.c_int
    cs = sc.sensors[2].c_struct()
    for flags in (S.EPSM_TRACE_GRADIENT_ONLY, S.EPSM_TRACE_GRADIENT_CAUSTIC):
        rc = fn(C.byref(sc.c_scene), C.byref(cs), C.c_uint32(1), 1, 4, 5, C.c_int64(0), C.c_int64(n), 0, C.c_void_p(ray[0].data_ptr()),
                C.c_void_p(ray[1].data_ptr()), C.c_void_p(ray[2].data_ptr()), C.c_void_p(ray[3].data_ptr()), None,
                C.c_void_p(rad.data_ptr()), None, None, C.c_uint32(flags), None)
        assert rc == -22


def test_zeroed_environment_is_no_environment_and_bad_tables_are_refused(scenes):
    """a zero-initialised EpsmScene.env means "no environment" (ABI 6), and every entry point refuses an
    environment whose tables are missing instead of faulting on the device."""
    import ctypes as C
    import copy
    from epsm_mitsuba3_amd import _lib, scene as S
    _, sc, _ = scenes[1]
    assert sc.c_scene.env.kind == 0
    lib = _lib.lib()
    dev = sc.device
    n = 64
    ray = torch.empty((4, n, 3), device=dev); rad = torch.empty((n, 3), device=dev)
    cs = sc.sensors[2].c_struct()
    bad = S.EpsmSceneC.from_buffer_copy(sc.c_scene)

    def call(scene_c):
        fn = lib.epsm_trace_paths
        fn.restype = C.c_int
        return fn(C.byref(scene_c), C.byref(cs), C.c_uint32(1), 1, 4, 5, C.c_int64(0), C.c_int64(n), 0, C.c_void_p(ray[0].data_ptr()),
                  C.c_void_p(ray[1].data_ptr()), C.c_void_p(ray[2].data_ptr()), C.c_void_p(ray[3].data_ptr()), None,
                  C.c_void_p(rad.data_ptr()), None, None, C.c_uint32(0), None)
    assert call(bad) == 0
    bad.env.kind, bad.env.emitter = 2, 0                      # an envmap without tables
    assert call(bad) == -22
    bad.env.kind, bad.env.emitter = 1, 99                     # a constant environment naming an emitter that does not exist
    assert call(bad) == -22
    bad.env.kind = 7
    assert call(bad) == -22
    bad.env.kind, bad.n_textures = 0, 3                       # textures announced, none given
    bad.textures = None
    assert call(bad) == -22
    torch.cuda.synchronize()


@pytest.mark.parametrize("variant", ["manifold", "manifold_caustic"])
def test_first_hit_fusion_gives_the_same_gradients(scenes, variant):
    """EPSM_TRACE_FUSE_FIRST_HIT (include/epsm_trace.h): the stage that shades a path's first hit does what the backward pass would do
    for a path WITHOUT a chain -- the first-vertex rows clamp(dldp) b_j of a diffuse first hit (epsm.py:250-272, 561-562, 791-792)
    and every path's share of d loss / d ray.o (:255-261) -- and such paths are not logged; the backward kernel is called without
    grad_o_sum and gives them no lane.  ``render_backward`` with and without it: the same buffers, the camera-origin gradient
    included (float order aside).  Scenes with the occluder record (max_depth <= 3) and launches the one-launch tracer takes are
    not fused and must agree trivially."""
    import epsm_mitsuba3_amd as epsm
    dev = torch.device("cuda", 0)
    fused_somewhere = 0
    for name, sc, max_depth in scenes:
        sc.tracer = "wavefront"
        res = sc.sensors[2].width
        g = torch.Generator().manual_seed(7)
        grad_in = (torch.randn((res, res, 5), generator=g) * 1e-3).to(dev)
        out = []
        for fuse in (False, True):
            integ = epsm.load_dict({"type": variant, "max_depth": max_depth, "fuse_first_hit": fuse})
            p = sc.param_grads()
            integ.render_backward(sc, p, grad_in, seed=11)
            torch.cuda.synchronize()
            assert bool(torch.isfinite(p.flat).all())
            out.append((p.flat.double().cpu(), p.cam_origin.double().cpu()))
        tiles = list(sc.iter_traces(sensor=2, seed=11, spp=integ.backward_spp, max_depth=integ.tracer_depth(), sparse_log=True,
                                    packed_log=True, gradient_only=variant,
                                    first_hit=(grad_in, sc.param_grads(), 0.1, True)))
        done = [t.log.first_hit_done for t in tiles]
        assert all(done) == (max_depth > 3), (name, done)                  # (the occluder record rides on the first vertex's emitter sample)
        if all(done):
            fused_somewhere += 1
            flags = torch.cat([t.log.flags for t in tiles])
            assert float((flags == 0).float().mean()) > (0.3 if variant == "manifold" else 0.1), name   # paths without a chain: most of a `manifold` wavefront
        m = float(out[0][0].abs().max())
        assert m > 0 or variant == "manifold_caustic", name
        assert float((out[0][0] - out[1][0]).abs().max()) <= 2e-5 * m + 1e-12, (name, m, float((out[0][0] - out[1][0]).abs().max()))
        # the camera-origin sum on its own scale (three numbers next to 10^5 rows)
        o0, o1 = out[0][1], out[1][1]
        assert float(o0.abs().max()) > 0, name
        assert float((o0 - o1).abs().max()) <= 2e-5 * float(o0.abs().max()) + 1e-12, (name, o0, o1)
    assert fused_somewhere >= 3
