"""GPU parity of epsm_first_vertex_tangent and epsm_scatter against the float64
restatements in oracle/epsm_oracle_aux.c (pinned by tests/test_tangent_scatter_oracle.py)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda", 0)


@pytest.mark.parametrize("res,spp,width", [(16, 8, 2), (33, 3, 12), (128, 8, 2)])
def test_tangent_matches_oracle(res, spp, width, dev):
    from epsm_mitsuba3_amd.synth import synth_camera_rays, synth_first_hit_triangles
    from epsm_mitsuba3_amd.tangent_scatter import first_vertex_tangent
    from oracle.binding import oracle_first_vertex_tangent
    o, d, dx, dy = synth_camera_rays(res, spp, seed=res)
    p0, p1, p2, b0, b1 = synth_first_hit_triangles(o, d, seed=res)
    g = torch.Generator().manual_seed(res)
    grad_in = torch.randn((res + 3, res + 5, 5), generator=g)
    active = torch.rand(d.shape[0], generator=g) > 0.1
    t_uv, t_p, t_o = oracle_first_vertex_tangent(o, d, dx, dy, grad_in, spp, res, p0, p1, p2, active, width)
    to = lambda t: t.to(dev)
    uv, p, go = first_vertex_tangent(to(o), to(d), to(dx), to(dy), to(grad_in), spp, res, to(p0), to(p1), to(p2),
                                     to(active), dlduv_width=width, want_origin_grad=True)
    assert uv.shape == t_uv.shape and p.shape == t_p.shape
    # fp32 closed form vs float64 dual numbers: tolerance 2e-4 of the per-path magnitude
    su = t_uv.abs().amax(dim=(1, 2)).clamp_min(1e-9)
    assert float(((uv.cpu().double() - t_uv).abs().amax(dim=(1, 2)) / su).max()) < 2e-4
    sp = t_p.abs().amax(dim=1).clamp_min(1e-9)
    assert float(((p.cpu().double() - t_p).abs().amax(dim=1) / sp).max()) < 2e-4
    inactive = ~active
    assert bool((uv.cpu()[inactive] == 0).all()) and bool((p.cpu()[inactive] == 0).all())
    if width > 2:
        assert bool((uv[:, :, 2:] == 0).all())
    assert torch.allclose(go.cpu().double(), t_o, rtol=1e-3, atol=1e-4 * float(t_o.abs().max() + 1))


@pytest.mark.parametrize("variant", ["manifold", "manifold_caustic"])
@pytest.mark.parametrize("V,coherent", [(7829, True), (100000, False)])
def test_scatter_matches_oracle(variant, V, coherent, dev):
    """Atomic order is free, so compare with the deterministic fp64 sum at 1e-4 of the
    buffer's magnitude (1e-5 relative per add, accumulated)."""
    import epsm_mitsuba3_amd as epsm
    from epsm_mitsuba3_amd.records import PackedRecords, PackedScatter
    from epsm_mitsuba3_amd.synth import synth_path_info, synth_scatter_info, path_info_to
    from epsm_mitsuba3_amd.tangent_scatter import scatter
    from epsm_mitsuba3_amd.manifold_grad import manifold_grad_packed
    from oracle.binding import oracle_scatter
    N, K, B = 30000, 4, 5
    profile = "mixed"
    pi, dlduv, dldp = synth_path_info(N, K, seed=9, profile=profile, tangent_scale=1e-4)
    si = synth_scatter_info(N, K, V, seed=9, n_bsdfs=B, coherent=coherent, shadow=True)      # + occluder record (epsm.py:609-620)
    rec = PackedRecords(path_info_to(pi, device=dev), device=dev)
    sc = PackedScatter(si, device=dev)
    out = manifold_grad_packed(variant, rec, dlduv.to(dev), dldp.to(dev), dlduv_cols=2)
    gp = torch.zeros((V, 3), device=dev); gn = torch.zeros((V, 3), device=dev); ga = torch.zeros(B, device=dev)
    scatter(variant, rec, sc, *out, gp, gn, ga)
    torch.cuda.synchronize()
    # the oracle scatters the SAME per-path gradients (the HIP outputs), so only the scatter is compared
    rp, rn, ra = oracle_scatter(variant, pi, si, out[0].cpu(), out[1].cpu(), out[2].cpu(), V, B)
    for mine, ref in ((gp, rp), (gn, rn), (ga, ra)):
        ref_max = float(ref.abs().max())
        assert ref_max > 0
        assert float((mine.cpu().double() - ref).abs().max()) <= 1e-4 * ref_max + 1e-9
    # accumulation semantics: a second call doubles the buffers
    scatter(variant, rec, sc, *out, gp, gn, ga)
    assert float((gp.cpu().double() - 2 * rp).abs().max()) <= 2e-4 * float(rp.abs().max()) + 1e-9
