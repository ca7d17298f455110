"""The tracer on the GPU: same records as the host build of the same per-path code (which the
analytic tests of tests/test_tracer_host.py pin), and the whole optimisation-step chain
trace -> tangent -> fused gradient/scatter on a real scene description."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _scene(device):
    from _scenes import floor_and_light
    sc = floor_and_light(res=32, bsdf={"type": "roughconductor", "material": "Al", "distribution": "ggx", "alpha": 0.1},
                         device=device)
    sc.attach("floor", positions=True)
    sc.attach_alpha("floor.bsdf")
    return sc


def test_gpu_records_match_host_build():
    from _scenes import on_host
    dev = torch.device("cuda", 0)
    g, h = _scene(dev), on_host(_scene("cpu"))
    n = 32 * 32 * 8
    a = g._trace(0, seed=3, spp=8, max_depth=4, K=3, lo=0, hi=n)
    b = h._trace(0, seed=3, spp=8, max_depth=4, K=3, lo=0, hi=n)
    assert torch.allclose(a.ray_d.cpu(), b.ray_d, atol=1e-6) and torch.allclose(a.film_pos.cpu(), b.film_pos, atol=1e-5)
    for k in (1, 2, 3):
        ra, rb = a.path_info[k], b.path_info[k]
        same = (ra["active"].cpu() == rb["active"]) & (ra["active_em"].cpu() == rb["active_em"]) & (ra["bsdf"].cpu() == rb["bsdf"])
        assert float(same.float().mean()) > 0.995, k           # fma contraction can flip a borderline decision
        for name in ("light", "hf", "eta"):
            x, y = ra[name].cpu()[same], rb[name][same]
            close = (x - y).abs().reshape(x.shape[0], -1).amax(dim=1) < 1e-3
            assert float(close.float().mean()) > 0.99, (k, name)
        x, y = ra["points"][3].cpu()[same], rb["points"][3][same]
        assert float(((x - y).abs().amax(dim=1) < 1e-3).float().mean()) > 0.99
    ta, tb = a.scatter_info[0]["tri"].cpu(), b.scatter_info[0]["tri"]
    assert float((ta == tb).float().mean()) > 0.995            # triangle ids
    assert torch.allclose(a.radiance.cpu().mean(0), b.radiance.mean(0), rtol=2e-2)


def test_primal_render_and_backward_on_a_scene():
    import epsm_mitsuba3_amd as epsm
    from epsm_mitsuba3_amd import scene as S
    from _scenes import quad, sensor
    dev = torch.device("cuda", 0)
    # a tilted specular plate under an area light above a diffuse floor: camera -> plate -> light
    pv, pf = quad(0.5, 0.8, up=True)
    pv = (S.rotate([1, 0, 0], 10.0)[:3, :3] @ pv.T).T
    fv, ff = quad(0.0, 4.0, up=True)
    lv, lf = quad(3.0, 0.5, up=False)
    cam = lambda res, spp: sensor([0, -3.0, 2.0], [0, 0, 0.4], up=(0, 0, 1), res=res, spp=spp, rfilter="gaussian")
    d = {"type": "scene", "s0": cam(64, 16), "s1": cam(64, 16), "s2": cam(32, 8),
         "plate": {"type": "mesh", "vertices": pv, "faces": pf,
                   "bsdf": {"type": "roughconductor", "material": "Al", "distribution": "ggx", "alpha": 0.05}},
         "floor": {"type": "mesh", "vertices": fv, "faces": ff, "face_normals": True, "bsdf": {"type": "diffuse"}},
         "light": {"type": "mesh", "vertices": lv, "faces": lf, "face_normals": True,
                   "emitter": {"type": "area", "radiance": {"type": "rgb", "value": 20.0}}}}
    sc = S.Scene.from_dict(d, device=dev)
    sc.attach("plate", positions=True, normals=True)
    integ = epsm.load_dict({"type": "manifold", "max_depth": 4})
    img = integ.render(sc, sensor=1, seed=0, spp=16)
    assert img.shape == (64, 64, 5) and bool((img[..., 3:] == 0).all())          # epsm.py:77-82
    assert float(img[..., :3].max()) > 1.0 and bool(torch.isfinite(img).all())
    params = sc.param_grads()
    g = torch.Generator().manual_seed(0)
    grad_in = torch.zeros((64, 64, 5)); grad_in[..., 3:] = torch.randn((64, 64, 2), generator=g) * 1e-2
    integ.render_backward(sc, params, grad_in.to(dev), sensor=1, seed=1, spp=16)
    torch.cuda.synchronize()
    gp = params.mesh_pos("plate")
    assert bool(torch.isfinite(params.flat).all()) and float(gp.abs().max()) > 0
    assert float(params.mesh_pos("floor").abs().max()) == 0                       # not attached
    # fused and two-stage agree on the real trace as well
    p2 = sc.param_grads()
    epsm.load_dict({"type": "manifold", "max_depth": 4, "fused": False}).render_backward(sc, p2, grad_in.to(dev), seed=1)
    m = float(p2.flat.abs().max())
    assert float((p2.flat - params.flat).abs().max()) <= 1e-3 * m
    # moving the plate changes the image (params.update())
    v = sc.vertex_positions("plate").clone(); v[:, 2] += 0.2
    sc.set_vertex_positions("plate", v)
    img2 = integ.render(sc, sensor=1, seed=0, spp=16)
    assert float((img2 - img).abs().mean()) > 1e-4


@pytest.mark.parametrize("tail", [True, False])
@pytest.mark.parametrize("max_depth,K", [(5, 5), (3, 2)])
def test_gpu_wavefront_tracer_equals_one_launch(max_depth, K, tail):
    """epsm_trace_paths_wavefront (queues of live paths, extend / shade / shadow kernels per bounce, ballot +
    prefix-count compaction) against epsm_trace_paths on the GPU: the same per-path code in another visiting order.
    The host builds of the two forms agree bit for bit (tests/test_tracer_wavefront_host.py); on the device the
    compiler may contract a product into an fma in one kernel and not in the other, so a borderline decision
    (hit / miss at a triangle's edge) may flip on a handful of paths.  ``tail``: with fewer than 2^19 paths alive the
    bounces >= 1 are ONE launch (wf_tail, csrc/epsm_trace_wavefront.h); EPSM_TRACE_NO_TAIL keeps the three stages."""
    from test_tracer_wavefront_host import _rich_scene, _all_arrays
    dev = torch.device("cuda", 0)
    res, spp = 48, 16
    sc = _rich_scene(res, spp, point_light=True, occluder=max_depth <= 3, device=dev)
    sc.wavefront_tail = tail
    n = res * res * spp
    sc.tracer = "mega"
    a = sc._trace(0, seed=5, spp=spp, max_depth=max_depth, K=K, lo=0, hi=n)
    sc.tracer = "wavefront"
    b = sc._trace(0, seed=5, spp=spp, max_depth=max_depth, K=K, lo=0, hi=n)
    torch.cuda.synchronize()
    x, y = _all_arrays(a), _all_arrays(b)
    assert x.keys() == y.keys()
    bad = torch.zeros(n, dtype=torch.bool, device=dev)
    for name in x:
        u, v = x[name].reshape(n, -1), y[name].reshape(n, -1)
        if name.endswith(".shadow") or name.endswith(".emit") or name.endswith(".aux"):
            # addressing records: integer words exactly, the float words (barycentrics, weights, d hf) to rounding
            fl = {"shadow": [1, 2, 3], "emit": [1, 2, 3], "aux": [1, 2, 3]}[name.rsplit(".", 1)[1]]
            it = [c for c in range(u.shape[1]) if c not in fl]
            uf, vf = u[:, fl].contiguous().view(torch.float32), v[:, fl].contiguous().view(torch.float32)
            differ = (u[:, it] != v[:, it]).any(dim=1) | ~(torch.isclose(uf, vf, rtol=1e-4, atol=1e-5) | (torch.isnan(uf) & torch.isnan(vf))).all(dim=1)
        elif u.dtype == torch.float32:
            differ = ~(torch.isclose(u, v, rtol=1e-4, atol=1e-5) | (torch.isnan(u) & torch.isnan(v))).all(dim=1)
        else:
            differ = (u != v).any(dim=1)
        if name in ("ray_o", "ray_d", "ray_dx", "ray_dy", "film_pos"):
            assert not bool(differ.any()), name
        bad |= differ
    assert float(bad.float().mean()) < 2e-3, f"{int(bad.sum())} of {n} paths differ"
    v1, v2 = a.path_info[1], a.path_info[2]
    assert 0 < int((v2["active"] > 0).sum()) < int((v1["active"] > 0).sum())
    if max_depth <= 3:
        sh_a, sh_b = a.scatter_info[0]["shadow"], b.scatter_info[0]["shadow"]
        assert int((sh_a[:, 0] != -1).sum()) > 100 and int((sh_b[:, 0] != -1).sum()) > 100
    assert torch.allclose(a.radiance.mean(0), b.radiance.mean(0), rtol=1e-3)


def test_gpu_wavefront_edge_sizes_and_empty_scene():
    """One path, a ragged handful, and a scene without a single triangle: the queues, chunk counts and the scan of
    the wavefront form at their smallest."""
    from test_tracer_wavefront_host import _rich_scene
    from _scenes import sensor
    from epsm_mitsuba3_amd import scene as S
    dev = torch.device("cuda", 0)
    sc = _rich_scene(8, 8, device=dev)
    for n, tail in ((1, True), (65, True), (257, True), (65, False), (257, False)):
        sc.wavefront_tail = tail
        sc.tracer = "mega"
        a = sc._trace(0, seed=3, spp=8, max_depth=4, K=3, lo=5, hi=5 + n)
        sc.tracer = "wavefront"
        b = sc._trace(0, seed=3, spp=8, max_depth=4, K=3, lo=5, hi=5 + n)
        torch.cuda.synchronize()
        assert torch.equal(a.path_info[1]["active"], b.path_info[1]["active"]) and torch.equal(a.valid, b.valid), n
        assert torch.allclose(a.radiance, b.radiance, rtol=1e-4, atol=1e-6), n
    empty = S.Scene.from_dict({"type": "scene", "cam": sensor([1.0, 2.0, 3.0], [1.0, 2.0, -5.0], up=(0, 1, 0), res=8, spp=2)}, device=dev)
    empty.tracer = "wavefront"
    t = empty._trace(0, seed=1, spp=2, max_depth=3, K=2, lo=0, hi=128)
    torch.cuda.synchronize()
    assert not bool(t.valid.any()) and float(t.radiance.abs().max()) == 0 and not bool(t.path_info[1]["active"].any())


def test_wavefront_workspace_is_checked():
    import ctypes as C
    from epsm_mitsuba3_amd import _lib
    lib = _lib.lib()
    assert lib.epsm_trace_workspace_bytes(C.c_int64(0)) == 0
    need = lib.epsm_trace_workspace_bytes(C.c_int64(1000))
    assert need >= 1000 * 172
    sc = _scene(torch.device("cuda", 0))
    sc.tracer = "wavefront"
    stream = torch.cuda.current_stream(torch.device("cuda", 0)).cuda_stream
    sc._wf_workspace[stream] = torch.empty(16, device="cuda", dtype=torch.uint8)       # too small: replaced by _trace
    tr = sc._trace(0, seed=1, spp=8, max_depth=3, K=2, lo=0, hi=500)
    torch.cuda.synchronize()
    assert sc._wf_workspace[stream].numel() >= lib.epsm_trace_workspace_bytes(C.c_int64(500)) and bool(torch.isfinite(tr.radiance).all())
    # a second stream gets its own scratch (two traces in flight must not share queues and path state)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        tr2 = sc._trace(0, seed=1, spp=8, max_depth=3, K=2, lo=0, hi=500)
    side.synchronize(); torch.cuda.synchronize()
    assert len(sc._wf_workspace) == 2 and torch.equal(tr.radiance, tr2.radiance)


@pytest.mark.parametrize("tracer", ["mega", "wavefront"])
def test_sparse_log_gives_the_same_gradients(tracer):
    """The backward pass on a sparse log (dead bounces carry only their mask fields, the rest of those rows is
    whatever the allocator handed out -- here NaNs on purpose) equals the backward pass on the dense log."""
    import epsm_mitsuba3_amd as epsm
    from test_tracer_wavefront_host import _rich_scene
    dev = torch.device("cuda", 0)
    res, spp = 32, 16
    sc = _rich_scene(res, spp, point_light=True, device=dev)
    sc.attach("floor", positions=True)
    sc.tracer = tracer
    n = res * res * spp
    g = torch.Generator().manual_seed(0)
    grad_in = torch.zeros((res, res, 5)); grad_in[..., 3:] = torch.randn((res, res, 2), generator=g) * 1e-2
    grad_in = grad_in.to(dev)
    for variant in ("manifold", "manifold_caustic"):
        integ = epsm.load_dict({"type": variant, "max_depth": 5})
        dense = sc._trace(0, seed=2, spp=spp, max_depth=5, K=5, lo=0, hi=n)
        pd = sc.param_grads()
        integ.backward_from_trace(dense, pd, grad_in)
        torch.cuda.synchronize()
        poison = [torch.full((n * 64,), float("nan"), device=dev) for _ in range(8)]   # what torch.empty will hand out next
        del poison
        sparse = sc._trace(0, seed=2, spp=spp, max_depth=5, K=5, lo=0, hi=n, sparse_log=True)
        ps = sc.param_grads()
        integ.backward_from_trace(sparse, ps, grad_in)
        torch.cuda.synchronize()
        m = float(pd.flat.abs().max())
        assert m > 0 and bool(torch.isfinite(ps.flat).all())
        assert float((pd.flat - ps.flat).abs().max()) <= 1e-4 * m, variant
        # two-stage route (calc_grad lists + scatter) on the sparse log as well
        p2 = sc.param_grads()
        epsm.load_dict({"type": variant, "max_depth": 5, "fused": False}).backward_from_trace(sparse, p2, grad_in)
        torch.cuda.synchronize()
        assert bool(torch.isfinite(p2.flat).all()) and float((pd.flat - p2.flat).abs().max()) <= 1e-3 * m, variant


@pytest.mark.parametrize("rfilter", [1, 0])                      # EPSM_RFILTER_GAUSSIAN, EPSM_RFILTER_BOX
@pytest.mark.parametrize("res,spp,n_cut", [(24, 64, 0), (24, 128, 0), (32, 16, 0), (40, 8, 0), (33, 5, 0), (24, 64, 37), (16, 0, 0)])
def test_film_splat_matches_host(res, spp, n_cut, rfilter):
    """epsm_film_splat against the serial host loop of tests/host_harness (ImageBlock::put restated twice,
    independently): pixel-major wavefronts at 64 / 128 / 16 / 8 spp take the wave-level sums over 64 / 16 / 8
    lanes, 5 spp, a ragged tail and positions all over the film (spp = 0 here) the lane-by-lane path."""
    import ctypes as C
    from _scenes import host_tracer
    from epsm_mitsuba3_amd import _lib
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(res * 1000 + spp)
    if spp:
        n = res * res * spp - n_cut
        pix = torch.arange(res * res * spp)[:n] // spp
        pos = torch.stack([(pix % res).float(), (pix // res).float()], dim=1) + torch.rand((n, 2), generator=g)
    else:
        n = 5000
        pos = torch.rand((n, 2), generator=g) * (res + 6) - 3.0           # some samples off the film
    rad = torch.rand((n, 3), generator=g) * 3.0
    want = torch.zeros((res, res, 4))
    assert host_tracer().epsm_film_splat(C.c_int64(n), C.c_void_p(pos.data_ptr()), C.c_void_p(rad.data_ptr()), res, res,
                                         rfilter, C.c_void_p(want.data_ptr()), None) == 0
    got = torch.zeros((res, res, 4), device=dev)
    pd, rd = pos.to(dev), rad.to(dev)
    assert _lib.lib().epsm_film_splat(C.c_int64(n), C.c_void_p(pd.data_ptr()), C.c_void_p(rd.data_ptr()), res, res, rfilter,
                                      C.c_void_p(got.data_ptr()), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)) == 0
    torch.cuda.synchronize()
    assert float(want.abs().max()) > 0
    assert torch.allclose(got.cpu(), want, rtol=2e-4, atol=2e-4 * float(want.abs().max()))


@pytest.mark.parametrize("tracer", ["mega", "wavefront"])
@pytest.mark.parametrize("max_depth", [5, 3])
def test_render_backward_on_the_native_packed_log(tracer, max_depth):
    """render_backward asks the tracer for the backward kernel's native layout (EPSM_TRACE_PACKED_LOG -> one launch of
    epsm_backward_pass_packed per tile) by default; ``packed_log=False`` keeps the reference's per-field tensors
    (-> epsm_backward_pass).  Same gradients -- positions, normals, alpha (its slot now comes from the triangle table),
    camera origin; NaN poison in fresh allocations shows that records of bounces a path never reached are not read."""
    import epsm_mitsuba3_amd as epsm
    from test_tracer_wavefront_host import _rich_scene
    dev = torch.device("cuda", 0)
    res, spp = 32, 16
    sc = _rich_scene(res, spp, point_light=True, occluder=max_depth <= 3, device=dev)
    sc.attach("floor", positions=True)
    sc.tracer = tracer
    sc.tile_paths = 5000                                   # several ragged tiles
    g = torch.Generator().manual_seed(0)
    grad_in = torch.zeros((res, res, 5)); grad_in[..., 3:] = torch.randn((res, res, 2), generator=g) * 1e-2
    grad_in = grad_in.to(dev)
    for variant in ("manifold", "manifold_caustic"):
        out = []
        # per-field tensors; the native log as two dense arrays (the default); as one interleaved block per path (ABI v7)
        for packed, layout in ((False, "dense"), (True, "dense"), (True, "interleaved")):
            sc.log_layout = layout
            integ = epsm.load_dict({"type": variant, "max_depth": max_depth, "packed_log": packed, "backward_sensor": 0})
            integ.backward_spp = spp
            poison = [torch.full((res * res * spp * 40,), float("nan"), device=dev) for _ in range(4)]
            del poison
            p = sc.param_grads()
            integ.render_backward(sc, p, grad_in, seed=2)
            torch.cuda.synchronize()
            assert bool(torch.isfinite(p.flat).all())
            out.append(p.flat.clone())
        m = float(out[0].abs().max())
        assert m > 0
        assert float((out[0] - out[1]).abs().max()) <= 2e-4 * m, variant
        assert float((out[0] - out[2]).abs().max()) <= 2e-4 * m, variant
    sc.log_layout = "dense"
    tiles = list(sc.iter_traces(sensor=0, seed=2, spp=spp, max_depth=max_depth, packed_log=True))
    assert len(tiles) == -(-res * res * spp // (max(5000, 1))) or sc.use_wavefront()
    assert all(t.log is not None and t.path_info is None and t.log.layout == "dense" for t in tiles)


def test_unlimited_depth_integrator_renders_and_differentiates():
    """``max_depth = -1`` (common.py:31-37: infinite) used to reach the tracer as 0xFFFFFFFF -> EINVAL.  The
    logging trace stops at 6 bounces whatever the integrator says (epsm.py:549); the primal pass goes as deep as the
    tracer does."""
    import epsm_mitsuba3_amd as epsm
    dev = torch.device("cuda", 0)
    sc = _scene(dev)
    integ = epsm.load_dict({"type": "manifold", "max_depth": -1})
    assert integ.tracer_depth() == 6 and "4294967295" in repr(integ)
    img = integ.render(sc, sensor=0, seed=1, spp=4)
    assert tuple(img.shape) == (32, 32, 5) and bool(torch.isfinite(img).all()) and float(img[..., :3].max()) > 0
    p = sc.param_grads()
    g = torch.zeros((32, 32, 5), device=dev); g[..., 3:] = 1e-2
    integ.backward_sensor = 0
    integ.render_backward(sc, p, g, seed=1)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(p.flat).all()) and float(p.flat.abs().max()) > 0


def test_gradient_image_does_not_depend_on_the_tile_size():
    """512 x 512 @ 64 spp on the 128 004-triangle scene: one tile of 2^24 paths (8.6 GB of packed log: 64-bit offsets
    everywhere) against sixteen tiles of 2^20 -- the same paths, the same sums up to the order of the atomics."""
    import epsm_mitsuba3_amd as epsm
    from epsm_mitsuba3_amd.exp import clutter
    dev = torch.device("cuda", 0)
    res, spp = 512, 64
    sc = clutter.load_scene(dev, n_spheres=100, res=res, spp=spp)
    for i in range(0, 100, 7):
        sc.attach(f"s{i}", positions=True, normals=True)
    sc.tracer = "wavefront"
    g = torch.Generator().manual_seed(4)
    grad_in = (torch.randn((res, res, 5), generator=g) * 1e-3).to(dev)
    integ = epsm.load_dict({"type": "manifold", "max_depth": clutter.max_depth})
    integ.backward_spp = spp
    out = []
    for tile in (1 << 24, 1 << 20):
        sc.WAVEFRONT_TILE_PATHS = tile
        assert len(list(t.path_offset for t in sc.iter_traces(sensor=2, seed=3, spp=spp, max_depth=clutter.max_depth, packed_log=True))) == (1 << 24) // tile
        p = sc.param_grads()
        integ.render_backward(sc, p, grad_in, seed=3)
        torch.cuda.synchronize()
        out.append(p.flat.double().cpu())
    m = float(out[0].abs().max())
    assert m > 0 and float((out[0] - out[1]).abs().max()) <= 2e-4 * m


@pytest.mark.parametrize("gradient_only", [False, True])
def test_tail_launch_gives_the_same_gradients(gradient_only):
    """The tail of the wavefront form (wf_tail: the paths still alive carried through the rest of their loop by one launch
    once fewer than 2^19 are left) against the three stages per bounce (EPSM_TRACE_NO_TAIL), at a size where the tail
    takes over in the MIDDLE of the loop: same parameter gradients from render_backward."""
    import epsm_mitsuba3_amd as epsm
    from epsm_mitsuba3_amd.exp import clutter
    dev = torch.device("cuda", 0)
    res, spp = 256, 16
    sc = clutter.load_scene(dev, n_spheres=40, res=res, spp=spp)
    sc.tracer = "wavefront"
    for i in range(0, 40, 3):
        sc.attach(f"s{i}", positions=True, normals=True)
    sc.attach("floor", positions=True)
    g = torch.Generator().manual_seed(3)
    grad_in = (torch.randn((res, res, 5), generator=g) * 1e-3).to(dev)
    out = []
    for tail in (False, True):
        sc.wavefront_tail = tail
        integ = epsm.load_dict({"type": "manifold", "max_depth": clutter.max_depth, "gradient_only": gradient_only})
        integ.backward_spp = spp
        p = sc.param_grads()
        integ.render_backward(sc, p, grad_in, seed=5)
        torch.cuda.synchronize()
        out.append(p.flat.double().cpu())
        if not tail:
            alive = sc.wavefront_queue_lengths()["alive"]
            if not gradient_only:                      # the tail starts at bounce 2 here, not at bounce 1
                assert alive[1] >= (1 << 19) > alive[2] > 0, alive
            else:
                assert (1 << 19) > alive[1] > alive[2] > 0, alive
    m = float(out[0].abs().max())
    assert m > 0
    assert float((out[0] - out[1]).abs().max()) <= 1e-5 * m, (m, float((out[0] - out[1]).abs().max()))
