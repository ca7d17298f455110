"""The whole backward pass on the GPU (tangent -> gradient -> scatter, per tile)
through the reference-shaped integrator surface, against the float64 oracle pipeline."""
import pytest
import torch

pytestmark = pytest.mark.gpu


# "pass": one launch per tile (epsm_backward_pass); "kernel": tangent kernel + fused gradient/scatter kernel;
# False: the reference's three stages (tangent, calc_grad lists, scatter)
@pytest.mark.usefixtures("window_form")
@pytest.mark.parametrize("fused", ["pass", "kernel", False])
@pytest.mark.parametrize("kind,profile", [("manifold", "bathroom"), ("manifold_caustic", "caustic"),
                                          ("manifold", "mixed"), ("manifold_caustic", "mixed")])
def test_render_backward_matches_oracle_pipeline(kind, profile, fused):
    import epsm_mitsuba3_amd as epsm
    from _pipeline_oracle import oracle_backward
    dev = torch.device("cuda", 0)
    res, spp, K, V, B = 32, 8, 4, 3000, 4       # 8 spp = the reference's backward wavefront (epsm.py:145)
    scene = epsm.SyntheticScene(res=res, n_vertices=K, n_scene_vertices=V, n_bsdfs=B, profile=profile,
                                device=dev, tile_paths=3000)
    integ = epsm.load_dict({"type": kind, "max_depth": 8, "fused": fused, "fuse_tangent": fused != "kernel"})
    integ.backward_spp = spp
    assert isinstance(integ, epsm.EPSMIntegrator) and integ.variant == kind and integ.fused == fused
    g = torch.Generator().manual_seed(11)
    grad_in = (torch.randn((res * 2, res * 2, 5), generator=g) * 1e-3).to(dev)   # tiled image; the crop is used
    params = epsm.ParamGrads(V, B, device=dev)
    integ.render_backward(scene, params, grad_in, sensor=1, seed=7, spp=64)      # sensor/spp ignored (epsm.py:142,145)
    torch.cuda.synchronize()
    traces = scene.trace_paths(seed=7, spp=integ.backward_spp)
    assert len(traces) == -(-res * res * spp // 3000) and traces[1].path_offset == 3000
    gp, gn, ga, go, allow = oracle_backward(kind, traces, grad_in.cpu(), V, B, straddle_band=0.02)
    for mine, ref, slack, name in ((params.pos, gp, allow[0], "pos"), (params.nrm, gn, allow[1], "nrm"),
                                   (params.alpha, ga, allow[2], "alpha"), (params.cam_origin, go, 0.0, "cam")):
        m = float(ref.abs().max())
        assert m > 0, name
        # end-to-end fp32 chain + atomic order: 2e-3 of the buffer's magnitude (as test_fused_matches_the_oracle_at_every_
        # coherence below), plus the weight of the per-path components within 2 % of the outlier threshold (fp32 and fp64
        # may zero different ones) and of the ill-conditioned paths on which the fp32 oracle itself leaves the fp64 one
        assert bool(((mine.cpu().double() - ref).abs() <= 2e-3 * m + slack).all()), (name, float(((mine.cpu().double() - ref).abs() - slack).max()) / m)
        if name in ("pos", "nrm"):
            assert float(slack.sum() / ref.abs().sum()) < 0.1, name      # the allowance is the exception, not the rule
    # accumulation: a second backward doubles the gradients (dr.backward accumulates)
    before = params.flat.clone()
    integ.render_backward(scene, params, grad_in, seed=7)
    assert torch.allclose(params.flat, 2 * before, rtol=1e-3, atol=1e-6 * float(before.abs().max()))


@pytest.mark.usefixtures("window_form")
@pytest.mark.parametrize("res,spp,V,max_depth", [(32, 8, 3000, 8), (12, 64, 300, 8), (16, 256, 120, 8), (32, 8, 3000, 3), (12, 64, 300, 3)])
@pytest.mark.parametrize("kind,profile", [("manifold", "bathroom"), ("manifold", "mixed"), ("manifold_caustic", "pool")])
def test_fused_matches_the_oracle_at_every_coherence(kind, profile, res, spp, V, max_depth):
    """The fused backward kernel (per-field records, tangents from the stand-alone kernel) against the float64 oracle
    pipeline, and against the reference's two stages on the GPU (dense calc_grad kernel + scatter kernel), at 8 / 64 /
    256 spp on finer and coarser meshes: at 64 / 256 spp on a coarse mesh a wave holds one first-hit triangle (every lane
    merges in the DPP sums) and a few per later bounce (leader rounds), at 8 spp nearly every row goes through the LDS
    table on its own.  max_depth = 3 adds the occluder record of the first vertex (epsm.py:609-620) to the trace.
    Against the oracle: 2e-3 of the buffer's magnitude plus the oracle's own allowance (components within 2 % of the
    outlier threshold, ill-conditioned paths: tests/_pipeline_oracle.py).  Against the two stages: tests/_util.py,
    two_routes_report."""
    import epsm_mitsuba3_amd as epsm
    from _pipeline_oracle import oracle_backward
    from _util import assert_two_routes_agree
    dev = torch.device("cuda", 0)
    K, B = 5, 4
    scene = epsm.SyntheticScene(res=res, n_vertices=K, n_scene_vertices=V, n_bsdfs=B, profile=profile,
                                device=dev, tile_paths=5000)
    g = torch.Generator().manual_seed(5)
    grad_in = (torch.randn((res, res, 5), generator=g) * 1e-3).to(dev)
    traces = scene.trace_paths(seed=3, spp=spp, max_depth=max_depth)
    assert (traces[0].scatter_info[0].get("shadow") is not None) == (max_depth <= 3)
    routes = {}
    for name, props in (("fused", {"fused": True, "fuse_tangent": False}), ("two stages", {"fused": False}),
                        ("lo", {"fused": False, "outlier_clip": 0.098}), ("hi", {"fused": False, "outlier_clip": 0.102})):
        integ = epsm.load_dict({"type": kind, "max_depth": 8, **props})
        params = epsm.ParamGrads(V, B, device=dev)
        for tr in traces:
            integ.backward_from_trace(tr, params, grad_in)
        torch.cuda.synchronize()
        routes[name] = params
    gp, gn, ga, go, allow = oracle_backward(kind, traces, grad_in.cpu(), V, B, straddle_band=0.02)
    for mine, ref, slack, name in ((routes["fused"].pos, gp, allow[0], "pos"), (routes["fused"].nrm, gn, allow[1], "nrm"),
                                   (routes["fused"].alpha, ga, allow[2], "alpha")):
        m = float(ref.abs().max())
        assert m > 0, name
        assert bool(((mine.cpu().double() - ref).abs() <= 2e-3 * m + slack).all()), name
    for attr in ("pos", "nrm", "alpha"):
        rep = assert_two_routes_agree(*(getattr(routes[k], attr).cpu() for k in ("fused", "two stages", "lo", "hi")),
                                      name=attr)
        print(kind, profile, res, spp, attr, rep)


@pytest.mark.usefixtures("window_form")
@pytest.mark.parametrize("K", [4, 5])
def test_fused_drops_the_rows_of_a_caustic_term_that_turns_non_finite(K):
    """ADVICE r1 / VERDICT r1: a ``manifold_caustic`` path whose solve at depth id* turns out non-finite contributes
    nothing to any parameter (the inverse is NaN -> nan_to_num, epsm.py:1076-1079).  The dense kernel zeroes the
    path's rows afterwards; the fused / one-launch kernels have already accumulated the rows of vertices 1..id*-2
    by then and must take them back (second turn of caustic_path, ScatterOut::undo_needed).  Shading normals along
    +x at vertex 3 or 4 (tangent 0/0 -> NaN, epsm.py:746-748) on half of the paths: fused == scatter(dense), and
    the poisoned paths contribute exactly nothing on the dense route."""
    import epsm_mitsuba3_amd as epsm
    from epsm_mitsuba3_amd.records import PackedRecords, PackedScatter
    from epsm_mitsuba3_amd.synth import synth_path_info, synth_scatter_info, path_info_to
    from epsm_mitsuba3_amd.tangent_scatter import manifold_grad_scatter
    from oracle.binding import oracle_scatter
    dev = torch.device("cuda", 0)
    n, V, B = 8192, 600, 4
    pi, dlduv, dldp = synth_path_info(n, K, seed=33, device=dev, profile="pool", p_terminate=0.0, p_not_mesh=0.0)
    si = synth_scatter_info(n, K, V, seed=33, device=dev, n_bsdfs=B, res=32, spp=8)
    bad3 = torch.arange(n, device=dev) % 4 == 1
    bad4 = torch.arange(n, device=dev) % 4 == 3
    for k, rows in ((3, bad3), (4, bad4)):
        if k <= K:
            for j in range(3):
                pi[k]["normals"][j][rows] = torch.tensor([1.0, 0.0, 0.0], device=dev)
    dense = epsm.calc_grad("manifold_caustic", pi, dlduv, dldp)
    torch.cuda.synchronize()
    fp = torch.stack(dense[0])
    assert bool(torch.isfinite(fp).all())
    # paths with a live term at or beyond the NaN vertex: every parameter row is exactly zero on the dense route,
    # and there are enough of them for the test to mean something
    clean = epsm.calc_grad("manifold_caustic", synth_path_info(n, K, seed=33, device=dev, profile="pool", p_terminate=0.0,
                                                              p_not_mesh=0.0)[0], dlduv, dldp)
    changed = (torch.stack(clean[0]) != fp).any(dim=2).any(dim=0)
    hit = changed & (bad3 | bad4)
    assert int(hit.sum()) > n // 20
    early = (torch.stack(clean[0])[:5] != 0).any(dim=2).any(dim=0)       # rows of vertex 1 (emitted before id* is reached)
    assert int((hit & early & (fp == 0).all(dim=2).all(dim=0)).sum()) > n // 50
    cpu_pi = path_info_to(pi, device="cpu")
    cpu_si = [{k: (v.cpu() if v is not None else None) for k, v in r.items()} for r in si]
    want = oracle_scatter("manifold_caustic", cpu_pi, cpu_si, *[[t.cpu() for t in lst] for lst in dense], V, B)
    gp, gn, ga = torch.zeros((V, 3), device=dev), torch.zeros((V, 3), device=dev), torch.zeros(B, device=dev)
    manifold_grad_scatter("manifold_caustic", PackedRecords(pi, device=dev), PackedScatter(si, device=dev), dlduv, dldp, gp, gn, ga)
    torch.cuda.synchronize()
    for mine, ref, name in ((gp, want[0], "pos"), (gn, want[1], "nrm"), (ga, want[2], "alpha")):
        m = float(ref.abs().max())
        assert m > 0, name
        assert float((mine.cpu().double() - ref).abs().max()) <= 2e-4 * m, name


def test_unknown_plugin_and_bad_props():
    import epsm_mitsuba3_amd as epsm
    with pytest.raises(RuntimeError):
        epsm.load_dict({"type": "manifold_shadow"})          # registered nowhere (EPSM/all.sh:8 would fail too)
    with pytest.raises(Exception):
        epsm.load_dict({"type": "manifold", "max_depth": -3})


@pytest.mark.parametrize("res,spp,V", [(512, 8, 50000), (128, 64, 2000)])
@pytest.mark.parametrize("kind,profile", [("manifold", "bathroom"), ("manifold", "specular"), ("manifold_caustic", "pool")])
def test_fused_equals_two_stage_at_scale(kind, profile, res, spp, V):
    """2^21 / 2^20 paths: the fused kernel and calc_grad -> scatter accumulate the same sums.  The two routes run
    different restatements of the per-path arithmetic (epsm_cp_core.h / epsm_path_core.h), so beyond the order of the
    float additions they may part where a component sits on the outlier threshold or a path is ill-conditioned:
    tests/_util.py, two_routes_report.  At 64 spp on a coarse mesh every wave of the fused kernel holds one or two
    triangles per bounce: its wave-level merge (DPP sums) does the adding."""
    import epsm_mitsuba3_amd as epsm
    from _util import assert_two_routes_agree
    dev = torch.device("cuda", 0)
    K, B = 5, 4
    scene = epsm.SyntheticScene(res=res, n_vertices=K, n_scene_vertices=V, n_bsdfs=B, profile=profile,
                                device=dev, tile_paths=res * res * spp)
    g = torch.Generator().manual_seed(2)
    grad_in = (torch.randn((res, res, 5), generator=g) * 1e-3).to(dev)
    bufs = []
    for props in ({"fused": True}, {"fused": False}, {"fused": False, "outlier_clip": 0.098}, {"fused": False, "outlier_clip": 0.102}):
        integ = epsm.load_dict({"type": kind, "max_depth": 8, **props})
        integ.backward_spp = spp
        params = epsm.ParamGrads(V, B, device=dev)
        integ.render_backward(scene, params, grad_in, seed=1)
        torch.cuda.synchronize()
        bufs.append(params.flat.double().cpu())
    print(kind, profile, res, spp, assert_two_routes_agree(*bufs))


@pytest.mark.parametrize("kind,profile,K,res,spp,V", [("manifold", "bathroom", 2, 256, 8, 7829), ("manifold_caustic", "pool", 4, 64, 32, 500),
                                                      ("manifold", "mixed", 5, 128, 16, 100000)])
def test_small_wavefront_forms_agree(kind, profile, K, res, spp, V):
    """2^17 .. 2^19 paths (the reference's own backward sizes): the routes a small wavefront can take -- small windows flushed
    into the library's replicas which the second, reducing kernel sums (the default), the same summed inside the launch by the
    last workgroup of each replica (EPSM_OPT_ONE_LAUNCH), without replicas, and the windows of 1024 paths of the large wavefronts -- and the reference's
    two stages accumulate the same sums, camera origin included; a second launch finds replicas and counters zeroed."""
    import epsm_mitsuba3_amd as epsm
    dev = torch.device("cuda", 0)
    B = 4
    scene = epsm.SyntheticScene(res=res, n_vertices=K, n_scene_vertices=V, n_bsdfs=B, profile=profile,
                                device=dev, tile_paths=res * res * spp)
    g = torch.Generator().manual_seed(5)
    grad_in = (torch.randn((res, res, 5), generator=g) * 1e-3).to(dev)
    bufs = {}
    from epsm_mitsuba3_amd import _lib
    default = dict(small_wavefront_paths=1 << 20, replicas=True, one_launch=False)         # (include/epsm.h, epsm_set_option)
    for name, opts, fused in (("replicas", {}, "pass"), ("replicas again", {}, "pass"), ("one launch", {"one_launch": True}, "pass"),
                              ("one launch again", {"one_launch": True}, "pass"), ("two launches after one", {}, "pass"),
                              ("direct", {"replicas": False}, "pass"),
                              ("windows of 1024", {"small_wavefront_paths": 0}, "pass"), ("two stages", {}, False),
                              ("lo", {}, False), ("hi", {}, False)):
        with _lib.options(**{**default, **opts}):
            integ = epsm.load_dict({"type": kind, "max_depth": 8, "fused": fused,
                                    "outlier_clip": {"lo": 0.098, "hi": 0.102}.get(name, 0.1)})
            integ.backward_spp = spp
            params = epsm.ParamGrads(V, B, device=dev)
            integ.render_backward(scene, params, grad_in, seed=1)
            torch.cuda.synchronize()
            bufs[name] = params.flat.double().cpu()
            if name == "replicas again":               # the workspace can be given back and comes back on demand
                assert _lib.lib().epsm_release_workspace() == 0
                params.zero_()
                integ.render_backward(scene, params, grad_in, seed=1)
                torch.cuda.synchronize()
                bufs["replicas after a release"] = params.flat.double().cpu()
    from _util import assert_two_routes_agree
    ref = bufs["replicas"]
    m = float(ref.abs().max())
    assert m > 0
    for name, b in bufs.items():
        if name in ("two stages", "lo", "hi"):
            continue
        assert float((b - ref).abs().max()) <= 2e-4 * m, name       # the same per-path arithmetic: order of the additions only
    # ... and the reference's two stages, whose dense kernel runs the other restatement of the per-path arithmetic
    print(kind, profile, assert_two_routes_agree(ref, bufs["two stages"], bufs["lo"], bufs["hi"]))


@pytest.mark.parametrize("kind,profile", [("manifold", "bathroom"), ("manifold_caustic", "pool")])
def test_tiny_terms_and_a_disabled_clamp_sum_alike_in_both_window_forms(kind, profile):
    """the accumulating kernel's LDS rows are 64-bit fixed point (44 fractional bits) while the clamp bounds the
    terms, float rows when the caller disables it (clip <= 0).  (a) A gradient image scaled to 1e-8 -- terms around 1e-11,
    as from a mean-normalised loss -- gives the sums of the float route (the reference-shaped two stages) in the small-
    wavefront form AND in the large one; (b) with the clamp off (float rows) the two forms agree with each other."""
    import epsm_mitsuba3_amd as epsm
    dev = torch.device("cuda", 0)
    res, spp, K, V, B = 128, 16, 4, 5000, 4
    scene = epsm.SyntheticScene(res=res, n_vertices=K, n_scene_vertices=V, n_bsdfs=B, profile=profile, device=dev, tile_paths=res * res * spp)
    g = torch.Generator().manual_seed(6)
    base = (torch.randn((res, res, 5), generator=g) * 1e-3).to(dev)
    for what, scale, clip in (("tiny terms", 1e-5, 0.1), ("clamp off", 1.0, 0.0)):
        bufs = {}
        from epsm_mitsuba3_amd import _lib
        for name, limit, fused in (("small form", 1 << 20, "pass"), ("large form", 0, "pass"), ("two stages", 1 << 20, False)):
            with _lib.options(small_wavefront_paths=limit):
                integ = epsm.load_dict({"type": kind, "max_depth": 8, "fused": fused, "outlier_clip": clip})
                integ.backward_spp = spp
                params = epsm.ParamGrads(V, B, device=dev)
                integ.render_backward(scene, params, base * scale, seed=2)
                torch.cuda.synchronize()
                bufs[name] = params.flat.double().cpu()
        ref = bufs["two stages"]
        m = float(ref.abs().max())
        assert m > 0 and torch.isfinite(ref).all(), what
        print(what, kind, "max |sum| of the two stages", m)
        for name in ("small form", "large form") if what == "tiny terms" else ():   # (unclamped, ill-conditioned paths dominate the two restatements' difference)
            d = (bufs[name] - ref).abs()
            # the two restatements of the per-path arithmetic part on ill-conditioned paths (test_small_wavefront_forms_agree);
            # what must NOT appear is a quantisation floor or a saturated row: mean difference and the bulk of the elements
            assert float(d.mean()) <= 2e-4 * m and float((d > 1e-2 * m).double().mean()) < 1e-3, (what, name, float(d.mean()) / m, float(d.max()) / m)
        ms = float(bufs["small form"].abs().max())
        assert torch.isfinite(bufs["small form"]).all() and torch.isfinite(bufs["large form"]).all(), what
        assert float((bufs["small form"] - bufs["large form"]).abs().max()) <= 2e-4 * ms, what     # same arithmetic: order of the additions only


def _permute_info(info, perm):
    out = []
    for rec in info:
        r = {}
        for k, v in rec.items():
            if k == "table":                       # the scene's triangle table is not a per-path array
                r[k] = v
            elif isinstance(v, (list, tuple)):
                r[k] = [x[perm] for x in v]
            elif isinstance(v, torch.Tensor):
                r[k] = v[perm]
            else:
                r[k] = v
        out.append(r)
    return out


@pytest.mark.parametrize("kind,profile,n", [("manifold", "bathroom", 1 << 22), ("manifold_caustic", "pool", 1 << 20),
                                            ("manifold", "mixed", (1 << 20) + 777)])
def test_fused_sums_do_not_depend_on_the_order_of_the_paths(kind, profile, n):
    """Size-independent property at 2^20 .. 2^22 paths: the parameter gradients are sums over paths, so any
    permutation of the wavefront gives the same buffers (up to the order of float additions).  The fused kernel
    regroups the paths of every 1024-path window by chain length and merges rows over waves: a shuffled
    wavefront runs through completely different windows, slots and merges."""
    from epsm_mitsuba3_amd.records import PackedRecords, PackedScatter
    from epsm_mitsuba3_amd.synth import synth_path_info, synth_scatter_info
    from epsm_mitsuba3_amd.tangent_scatter import manifold_grad_scatter
    dev = torch.device("cuda", 0)
    K, V, B = 5, 20000, 4
    pi, dlduv, dldp = synth_path_info(n, K, seed=21, device=dev, profile=profile)
    si = synth_scatter_info(n, K, V, seed=21, device=dev, n_bsdfs=B, res=256, spp=64, shadow=True)
    g = torch.Generator(device=dev).manual_seed(4)
    perm = torch.randperm(n, generator=g, device=dev)
    bufs = []
    for p in (None, perm):
        a, b = (pi, si) if p is None else (_permute_info(pi, p), _permute_info(si, p))
        d, q = (dlduv, dldp) if p is None else (dlduv[p], dldp[p])
        gp, gn, ga = torch.zeros((V, 3), device=dev), torch.zeros((V, 3), device=dev), torch.zeros(B, device=dev)
        manifold_grad_scatter(kind, PackedRecords(a, device=dev), PackedScatter(b, device=dev), d, q, gp, gn, ga)
        torch.cuda.synchronize()
        bufs.append([t.double().cpu() for t in (gp, gn, ga)])
    for x, y, name in zip(bufs[0], bufs[1], ("pos", "nrm", "alpha")):
        m = float(x.abs().max())
        assert m > 0, name
        assert float((x - y).abs().max()) <= 3e-4 * m, name


@pytest.mark.usefixtures("window_form")
@pytest.mark.parametrize("kind,profile", [("manifold", "bathroom"), ("manifold", "specular"), ("manifold_caustic", "pool")])
def test_two_routes_agree_tightly_on_well_conditioned_paths(kind, profile):
    """the comparison of the two GPU routes elsewhere (tests/_util.py, two_routes_report) makes room for components on
    the +-0.1 threshold and for ill-conditioned paths; a dropped row or a mis-keyed one on a FEW rows would hide in that room.
    Here neither exists: the paths are the well-conditioned ones of a synthetic wavefront (cond_2 < 100 of every system they
    use, from the float64 oracle) and the tangents are small enough that no component comes near the clamp -- so the
    accumulating kernel (csrc/epsm_cp_core.h) and the reference-shaped two stages (csrc/epsm_path_core.h + the scatter kernel)
    must agree to 2e-4 of the buffer's magnitude in the MAXIMUM norm, every buffer."""
    import epsm_mitsuba3_amd as epsm
    from _util import select_paths
    from epsm_mitsuba3_amd.records import PackedRecords, PackedScatter
    from epsm_mitsuba3_amd.synth import path_info_to
    from epsm_mitsuba3_amd.tangent_scatter import manifold_grad_scatter, scatter
    from oracle.binding import oracle_cond
    dev = torch.device("cuda", 0)
    res, spp, K, V, B = 50, 8, 5, 3000, 4
    N = res * res * spp
    scene = epsm.SyntheticScene(res=res, n_vertices=K, n_scene_vertices=V, n_bsdfs=B, profile=profile, device=dev, tile_paths=N)
    trace = scene.tile(0, 0, N, seed=21, spp=spp, K=K)
    g = torch.Generator().manual_seed(8)
    dlduv = torch.zeros((N, 1, 2 * (K + 1)))
    dlduv[:, 0, :2] = torch.randn((N, 2), generator=g) * 2e-5
    dldp = torch.randn((N, 3), generator=g) * 2e-5
    cond = oracle_cond(kind, path_info_to(trace.path_info, device="cpu"), dlduv.double(), dldp.double())
    idx = torch.nonzero(cond < 100.0).flatten()
    assert idx.numel() > 0.25 * N
    sub = select_paths(trace, idx)
    d, q = dlduv[idx][:, :, :2].contiguous().to(dev), dldp[idx].contiguous().to(dev)
    rec, sc = PackedRecords(sub.path_info, device=dev), PackedScatter(sub.scatter_info, device=dev)
    fused = [torch.zeros((V, 3), device=dev), torch.zeros((V, 3), device=dev), torch.zeros(B, device=dev)]
    manifold_grad_scatter(kind, rec, sc, d, q, *fused)
    from epsm_mitsuba3_amd.manifold_grad import manifold_grad_packed
    fp, lg, dg = manifold_grad_packed(kind, rec, d, q, dlduv_cols=2)
    assert max(float(t.abs().max()) for t in (fp, lg, dg)) < 0.09    # nothing near the clamp
    two = [torch.zeros((V, 3), device=dev), torch.zeros((V, 3), device=dev), torch.zeros(B, device=dev)]
    scatter(kind, rec, sc, fp, lg, dg, *two)
    torch.cuda.synchronize()
    for a, b, name in zip(fused, two, ("pos", "nrm", "alpha")):
        m = float(b.abs().max())
        assert m > 0 or name == "alpha", name
        assert float((a - b).abs().max()) <= 2e-4 * m + 1e-30, (name, float((a - b).abs().max()) / max(m, 1e-30))


def test_full_size_wavefront_of_config_2():
    """BASELINE.json configs[1] at full size -- 512 x 512 @ 64 spp = 16 777 216 paths, K = 5, the wavefront bench.py
    times -- through size-independent properties (the oracle is too slow there): the one-launch backward pass and
    the reference-shaped three stages (tangent -> calc_grad lists -> scatter) are two independent routes to the same
    sums over paths; two runs of the one-launch route differ only by the order of float additions; the camera-origin
    gradient is minus the sum of the ray-direction tangents whichever kernel sums it."""
    import epsm_mitsuba3_amd as epsm
    from epsm_mitsuba3_amd.records import PackedRecords, PackedScatter, num_param_grads
    dev = torch.device("cuda", 0)
    res, spp, K, V, B = 512, 64, 5, 100000, 4
    N = res * res * spp
    scene = epsm.SyntheticScene(res=res, n_vertices=K, n_scene_vertices=V, n_bsdfs=B, profile="bathroom", device=dev, tile_paths=N)
    trace = scene.tile(0, 0, N, seed=0, spp=spp, K=K)
    packed = (PackedRecords(trace.path_info, device=dev), PackedScatter(trace.scatter_info, device=dev))
    P = num_param_grads("manifold", K)
    out = (torch.empty((P, N, 3), device=dev), torch.empty((K, N, 3), device=dev), torch.empty((K, N, 3), device=dev))
    g = torch.Generator(device=dev).manual_seed(1)
    grad_in = torch.randn((res, res, 5), generator=g, device=dev) * 1e-3
    bufs = []
    for cfg in ({"fused": True}, {"fused": True}, {"fused": False}, {"fused": False, "outlier_clip": 0.098},
                {"fused": False, "outlier_clip": 0.102}):
        integ = epsm.load_dict({"type": "manifold", "max_depth": 8, **cfg})
        params = epsm.ParamGrads(V, B, device=dev)
        integ.backward_from_trace(trace, params, grad_in, packed=packed, out=out)
        torch.cuda.synchronize()
        assert bool(torch.isfinite(params.flat).all())
        bufs.append(params.flat.double().cpu())
    m = float(bufs[0].abs().max())
    assert m > 0
    from _util import assert_two_routes_agree
    assert float((bufs[0] - bufs[1]).abs().max()) <= 1e-5 * m                   # run to run: order of the atomics only
    print(assert_two_routes_agree(bufs[0], bufs[2], bufs[3], bufs[4]))          # one launch vs three stages (tests/_util.py)
    assert abs(float(bufs[0].sum() - bufs[2].sum())) <= 1e-4 * float(bufs[0].abs().sum()) + float((bufs[3] - bufs[4]).abs().sum())   # checksum of the whole buffer


@pytest.mark.usefixtures("window_form")
@pytest.mark.parametrize("K", [1, 2, 3, 4, 5])
@pytest.mark.parametrize("kind,profile,max_depth", [("manifold", "bathroom", 8), ("manifold", "mixed", 3),
                                                    ("manifold_caustic", "pool", 8), ("manifold_caustic", "mixed", 3)])
def test_packed_log_gives_the_sums_of_the_per_array_records(kind, profile, K, max_depth):
    """The native log (EpsmPackedLog: one 128-byte record per path vertex, rays (N,12), one flag word per path) through
    ``epsm_backward_pass_packed`` against the same trace in the reference's tensor layout through ``epsm_backward_pass``:
    the same sums (float order aside), camera-origin gradient included; max_depth = 3 adds the occluder record."""
    import epsm_mitsuba3_amd as epsm
    from epsm_mitsuba3_amd.records import PackedLog
    dev = torch.device("cuda", 0)
    res, spp, V, B = 24, 16, 900, 3
    scene = epsm.SyntheticScene(res=res, n_vertices=K, n_scene_vertices=V, n_bsdfs=B, profile=profile, device=dev,
                                tile_paths=res * res * spp)
    g = torch.Generator().manual_seed(6)
    grad_in = (torch.randn((res, res, 5), generator=g) * 1e-3).to(dev)
    integ = epsm.load_dict({"type": kind, "max_depth": max_depth})
    (trace,) = scene.trace_paths(seed=4, spp=spp, max_depth=max_depth)
    a, b = epsm.ParamGrads(V, B, device=dev), epsm.ParamGrads(V, B, device=dev)
    integ.backward_from_trace(trace, a, grad_in)
    # both placements of the log (ABI v7): one interleaved block of K + 1 cache lines per path, or two dense arrays
    for layout in ("interleaved", "dense"):
        log = PackedLog.from_trace(trace, layout=layout)
        assert log.layout == layout and (log.shadow is not None) == (max_depth <= 3)
        assert log.rays.data_ptr() % 128 == 0 and (layout == "dense" or log.verts.data_ptr() % 128 == 64)
        b.flat.zero_()
        integ.backward_from_trace(trace, b, grad_in, packed=log)
        torch.cuda.synchronize()
        for x, y, name in ((a.pos, b.pos, "pos"), (a.nrm, b.nrm, "nrm"), (a.alpha, b.alpha, "alpha"), (a.cam_origin, b.cam_origin, "cam")):
            m = float(x.abs().max())
            if name in ("pos", "cam"):
                assert m > 0, name
            assert float((x - y).abs().max()) <= 2e-4 * m + 1e-12, (name, layout)


@pytest.mark.parametrize("res,spp", [(24, 16), (64, 512)])          # a small wavefront (one window per workgroup) and a large one (windows of 2048)
@pytest.mark.parametrize("kind,profile", [("manifold", "bathroom"), ("manifold_caustic", "pool")])
def test_path_list_gives_the_sums_of_the_whole_log(kind, profile, res, spp):
    """EpsmPackedLog.path_list / path_count (include/epsm.h): the backward pass runs its windows over a LIST of paths -- what the tracer
    hands over under EPSM_TRACE_FUSE_FIRST_HIT, every other path's flag word being 0 -- with the count read on the device.  A
    synthetic log with 85 % of its flag words zeroed: the launch over the list of the others against the launch over all N (both
    without the camera-origin sum, which the list form does not take): the same rows."""
    import epsm_mitsuba3_amd as epsm
    from epsm_mitsuba3_amd.records import PackedLog
    from epsm_mitsuba3_amd.tangent_scatter import backward_pass_packed
    dev = torch.device("cuda", 0)
    K, V, B = 4, 3000, 3
    N = res * res * spp
    scene = epsm.SyntheticScene(res=res, n_vertices=K, n_scene_vertices=V, n_bsdfs=B, profile=profile, device=dev, tile_paths=N)
    g = torch.Generator().manual_seed(9)
    grad_in = (torch.randn((res, res, 5), generator=g) * 1e-3).to(dev)
    (trace,) = scene.trace_paths(seed=4, spp=spp, max_depth=8)
    log = PackedLog.from_trace(trace)
    keep = torch.rand(N, generator=g) < 0.15
    keep[:5] = True; keep[-3:] = True
    log.flags[~keep.to(dev)] = 0
    out = []
    for use_list in (False, True):
        if use_list:
            lst = torch.nonzero(keep).flatten().to(torch.int32).to(dev)
            cap = torch.zeros(N, dtype=torch.int32, device=dev); cap[:lst.numel()] = lst          # (capacity N, as the tracer allocates it)
            log.set_path_list(cap, torch.tensor([lst.numel()], dtype=torch.int32, device=dev))
        p = epsm.ParamGrads(V, B, device=dev)
        backward_pass_packed(kind, log, grad_in, spp, res, p.pos, p.nrm, p.alpha, None, clip=0.1, path_offset=0)
        torch.cuda.synchronize()
        out.append(p.flat.double().cpu())
    m = float(out[0].abs().max())
    assert m > 0 and float((out[0] - out[1]).abs().max()) <= 1e-5 * m
    with pytest.raises(Exception):                                   # the list form takes no grad_o_sum
        backward_pass_packed(kind, log, grad_in, spp, res, p.pos, p.nrm, p.alpha, p.cam_origin, clip=0.1, path_offset=0)
