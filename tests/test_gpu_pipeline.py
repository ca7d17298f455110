"""The whole backward pass on the GPU (tangent -> gradient -> scatter, per tile)
through the reference-shaped integrator surface, against the float64 oracle pipeline."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("kind,profile", [("manifold", "bathroom"), ("manifold_caustic", "caustic"),
                                          ("manifold", "mixed"), ("manifold_caustic", "mixed")])
def test_render_backward_matches_oracle_pipeline(kind, profile, fused):
    import epsm_mitsuba3_amd as epsm
    from _pipeline_oracle import oracle_backward
    dev = torch.device("cuda", 0)
    res, spp, K, V, B = 32, 8, 4, 3000, 4
    scene = epsm.SyntheticScene(res=res, n_vertices=K, n_scene_vertices=V, n_bsdfs=B, profile=profile,
                                device=dev, tile_paths=3000)
    integ = epsm.load_dict({"type": kind, "max_depth": 8, "fused": fused})
    assert isinstance(integ, epsm.EPSMIntegrator) and integ.variant == kind and integ.fused == fused
    g = torch.Generator().manual_seed(11)
    grad_in = (torch.randn((res * 2, res * 2, 5), generator=g) * 1e-3).to(dev)   # tiled image; the crop is used
    params = epsm.ParamGrads(V, B, device=dev)
    integ.render_backward(scene, params, grad_in, sensor=1, seed=7, spp=64)      # sensor/spp ignored (epsm.py:142,145)
    torch.cuda.synchronize()
    traces = scene.trace_paths(seed=7, spp=integ.backward_spp)
    assert len(traces) == 3 and traces[1].path_offset == 3000
    gp, gn, ga, go = oracle_backward(kind, traces, grad_in.cpu(), V, B)
    for mine, ref, name in ((params.pos, gp, "pos"), (params.nrm, gn, "nrm"), (params.alpha, ga, "alpha"),
                            (params.cam_origin, go, "cam")):
        m = float(ref.abs().max())
        assert m > 0, name
        # end-to-end fp32 chain + clamp straddlers + atomic order: 1 % of the buffer's magnitude
        assert float((mine.cpu().double() - ref).abs().max()) <= 1e-2 * m, name
    # accumulation: a second backward doubles the gradients (dr.backward accumulates)
    before = params.flat.clone()
    integ.render_backward(scene, params, grad_in, seed=7)
    assert torch.allclose(params.flat, 2 * before, rtol=1e-3, atol=1e-6 * float(before.abs().max()))


def test_unknown_plugin_and_bad_props():
    import epsm_mitsuba3_amd as epsm
    with pytest.raises(RuntimeError):
        epsm.load_dict({"type": "manifold_shadow"})          # registered nowhere (EPSM/all.sh:8 would fail too)
    with pytest.raises(Exception):
        epsm.load_dict({"type": "manifold", "max_depth": -3})


@pytest.mark.parametrize("kind,profile", [("manifold", "bathroom"), ("manifold", "specular"), ("manifold_caustic", "pool")])
def test_fused_equals_two_stage_at_scale(kind, profile):
    """2^21 paths: the fused kernel and calc_grad -> scatter accumulate the same sums (the
    only difference is the order of float additions)."""
    import epsm_mitsuba3_amd as epsm
    dev = torch.device("cuda", 0)
    res, spp, K, V, B = 512, 8, 5, 50000, 4
    scene = epsm.SyntheticScene(res=res, n_vertices=K, n_scene_vertices=V, n_bsdfs=B, profile=profile,
                                device=dev, tile_paths=res * res * spp)
    g = torch.Generator().manual_seed(2)
    grad_in = (torch.randn((res, res, 5), generator=g) * 1e-3).to(dev)
    bufs = []
    for fused in (True, False):
        integ = epsm.load_dict({"type": kind, "max_depth": 8, "fused": fused})
        params = epsm.ParamGrads(V, B, device=dev)
        integ.render_backward(scene, params, grad_in, seed=1)
        torch.cuda.synchronize()
        bufs.append(params.flat.double().cpu())
    m = float(bufs[1].abs().max())
    assert m > 0
    assert float((bufs[0] - bufs[1]).abs().max()) <= 2e-4 * m
