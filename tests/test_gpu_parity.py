"""GPU parity tests: the HIP path (through the C ABI) against the golden vectors
of the reference, against the oracle on seeded synthetic records, and through
size-independent properties at the full benchmark size."""
import pytest
import torch

from _util import golden_files, golden_id, load_golden, stack3, parity_report, gated_parity_report

pytestmark = pytest.mark.gpu

FILES = golden_files()


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda", 0)


def _gpu(pi, dev):
    from epsm_mitsuba3_amd.synth import path_info_to
    return path_info_to(pi, device=dev)


@pytest.mark.parametrize("path", FILES, ids=golden_id)
def test_hip_matches_reference_golden(path, dev):
    """Tolerance.  Inside the conditioning gate of SURVEY.md 8c (cond_2 < 1e4 of the systems the path uses, from the
    oracle): error <= scale * max(2e-4, 4 eps32 cond) on all but 0.5 % of the paths (one path on the 96..128-path files,
    five on the 1024-path ones), no yardstick.  The small files are additionally held to the reference's own fp32 run as a
    yardstick -- per path, max-norm error <= 2e-4*scale + 16x the reference's fp32 distance to its float64 run -- with at
    most ONE path beyond it."""
    import epsm_mitsuba3_amd as epsm
    variant, pi, dlduv, dldp, ref = load_golden(path)
    fp, lg, dg = epsm.calc_grad(variant, _gpu(pi, dev), dlduv.to(dev), dldp.to(dev))
    assert len(fp) == ref["ref32_param"].shape[0] and len(lg) == len(dg) == len(pi) - 1
    mine = stack3(fp, lg, dg)
    truth = torch.cat([ref["ref64_param"], ref["ref64_light"], ref["ref64_diffuse"]]).double()
    yard = torch.cat([ref["ref32_param"], ref["ref32_light"], ref["ref32_diffuse"]]).double()
    assert not torch.isnan(mine).any()
    rep = parity_report(mine, truth, yard)
    from oracle.binding import oracle_cond
    gated = gated_parity_report(mine, truth, oracle_cond(variant, pi, dlduv, dldp))
    print(golden_id(path), "yardstick:", rep, "gated:", gated)
    if truth.shape[1] < 1024:                       # the 96..128-path files: one path is already 1 %
        assert rep["n_bad"] <= 1, rep
    # (the three 1024-path files -- the benchmark profiles -- are held to the conditioning gate alone: no yardstick)
    assert rep["median_rel"] < 1e-4, rep
    assert gated["n_bad_inside"] <= max(1, int(0.005 * truth.shape[1])), gated
    # masked paths must be EXACT zeros: wherever the float64 reference is
    # zero for a whole path, so are we
    dead = (truth == 0).all(dim=2).all(dim=0)
    assert bool((mine[:, dead] == 0).all())


@pytest.mark.parametrize("variant,profile", [("manifold", "bathroom"), ("manifold", "specular"),
                                              ("manifold", "mixed"), ("manifold_caustic", "caustic"),
                                              ("manifold_caustic", "pool"), ("manifold_caustic", "mixed")])
@pytest.mark.parametrize("K", [1, 2, 3, 4, 5])
def test_hip_matches_oracle_synthetic(variant, profile, K, dev):
    import epsm_mitsuba3_amd as epsm
    from epsm_mitsuba3_amd.synth import synth_path_info
    from oracle.binding import oracle_calc_grad, oracle_cond
    N = 20000
    pi, dlduv, dldp = synth_path_info(N, K, seed=40 + K, profile=profile, tangent_scale=2e-5)
    fp, lg, dg = epsm.calc_grad(variant, _gpu(pi, dev), dlduv.to(dev), dldp.to(dev))
    t = oracle_calc_grad(variant, pi, dlduv, dldp, dtype=torch.float64)
    y = oracle_calc_grad(variant, pi, dlduv, dldp, dtype=torch.float32)
    rep = parity_report(stack3(fp, lg, dg), stack3(*t[:3]), stack3(*y[:3]))
    assert rep["frac_bad"] <= 0.005, rep
    assert rep["median_rel"] < 1e-4, rep
    # inside cond_2 < 1e4 at most 0.5 % of the paths beyond scale * max(2e-4, 4 eps cond), no yardstick;
    # the fraction outside the gate is reported (pytest -s), not bounded
    gated = gated_parity_report(stack3(fp, lg, dg), stack3(*t[:3]), oracle_cond(variant, pi, dlduv, dldp))
    print(variant, profile, K, gated)
    assert gated["frac_bad_inside"] <= 0.005, gated


def test_ragged_and_tiny_sizes(dev):
    import epsm_mitsuba3_amd as epsm
    from epsm_mitsuba3_amd.synth import synth_path_info
    from oracle.binding import oracle_calc_grad
    for N in (1, 63, 64, 65, 255, 257, 1000):
        pi, dlduv, dldp = synth_path_info(N, 3, seed=N, profile="mixed", tangent_scale=1e-5)
        for variant in ("manifold", "manifold_caustic"):
            fp, lg, dg = epsm.calc_grad(variant, _gpu(pi, dev), dlduv.to(dev), dldp.to(dev))
            t = oracle_calc_grad(variant, pi, dlduv, dldp, dtype=torch.float64)
            rep = parity_report(stack3(fp, lg, dg), stack3(*t[:3]), None, rel=5e-3)
            assert rep["n_bad"] <= max(1, N // 50), (N, variant, rep)


def test_empty_wavefront(dev):
    import epsm_mitsuba3_amd as epsm
    from epsm_mitsuba3_amd.synth import synth_path_info
    pi, dlduv, dldp = synth_path_info(0, 2, seed=0)
    fp, lg, dg = epsm.calc_grad("manifold", _gpu(pi, dev), dlduv.to(dev), dldp.to(dev))
    assert len(fp) == 10 and all(x.shape == (0, 3) for x in fp + lg + dg)


def test_general_dlduv_columns(dev):
    """calc_grad accepts tangents on every vertex's barycentrics (epsm.py:850-851 uses
    dlduv[..., :2id]); render_backward only fills the first two."""
    import epsm_mitsuba3_amd as epsm
    from epsm_mitsuba3_amd.synth import synth_path_info
    from oracle.binding import oracle_calc_grad
    pi, dlduv, dldp = synth_path_info(5000, 4, seed=5, profile="mixed", tangent_scale=1e-5)
    g = torch.Generator().manual_seed(1)
    dlduv[:, 0, :] = torch.randn(dlduv.shape[0], dlduv.shape[2], generator=g) * 1e-5
    for variant in ("manifold", "manifold_caustic"):
        fp, lg, dg = epsm.calc_grad(variant, _gpu(pi, dev), dlduv.to(dev), dldp.to(dev))
        t = oracle_calc_grad(variant, pi, dlduv, dldp, dtype=torch.float64)
        y = oracle_calc_grad(variant, pi, dlduv, dldp, dtype=torch.float32)
        rep = parity_report(stack3(fp, lg, dg), stack3(*t[:3]), stack3(*y[:3]))
        assert rep["frac_bad"] <= 0.01, (variant, rep)


@pytest.mark.parametrize("variant,profile", [("manifold", "bathroom"), ("manifold_caustic", "pool")])
def test_full_size_properties(variant, profile, dev):
    """Size-independent properties on a 2^24-path wavefront -- BASELINE.json configs[1] at full size (oracle too slow there):
    determinism, exact linearity in the tangents (scaling by 2 is exact in binary
    floating point when the clamp is off), permutation equivariance, and a spot
    check of 4096 random paths against the oracle."""
    from epsm_mitsuba3_amd.synth import synth_path_info
    from epsm_mitsuba3_amd.records import PackedRecords
    from epsm_mitsuba3_amd.manifold_grad import manifold_grad_packed
    from oracle.binding import oracle_calc_grad
    N, K = 1 << 24, 5
    pi, dlduv, dldp = synth_path_info(N, K, seed=2, device=dev, profile=profile, tangent_scale=1e-5)
    rec = PackedRecords(pi, device=dev)
    a = manifold_grad_packed(variant, rec, dlduv, dldp, clip=0.0, dlduv_cols=2)
    b = manifold_grad_packed(variant, rec, dlduv, dldp, clip=0.0, dlduv_cols=2)
    for x, y in zip(a, b):
        assert torch.equal(x, y), "not deterministic"
    c = manifold_grad_packed(variant, rec, dlduv * 2, dldp * 2, clip=0.0, dlduv_cols=2)
    for x, y in zip(a, c):
        assert torch.equal(x * 2, y), "not linear in the tangents"
    # clamp: the clipped result equals the unclipped one with |g| > 0.1 zeroed (epsm.py:932-944)
    big = manifold_grad_packed(variant, rec, dlduv * 64, dldp * 64, clip=0.0, dlduv_cols=2)
    cl = manifold_grad_packed(variant, rec, dlduv * 64, dldp * 64, clip=0.1, dlduv_cols=2)
    for x, y in zip(big, cl):
        assert torch.equal(torch.where(x.abs() > 0.1, torch.zeros_like(x), x), y)
        assert not torch.isnan(y).any()
    # spot check against the oracle
    idx = torch.randperm(N, device=dev)[:4096]
    sub = []
    for r in pi:
        q = {}
        for k, v in r.items():
            if isinstance(v, (list, tuple)):
                q[k] = [t[idx].cpu() for t in v]
            elif isinstance(v, torch.Tensor):
                q[k] = v[idx].cpu()
            else:
                q[k] = v
        sub.append(q)
    cl1 = manifold_grad_packed(variant, rec, dlduv, dldp, clip=0.1, dlduv_cols=2)
    mine = torch.cat([t[:, idx].cpu().double() for t in cl1])
    t64 = oracle_calc_grad(variant, sub, dlduv[idx].cpu(), dldp[idx].cpu(), dtype=torch.float64)
    y32 = oracle_calc_grad(variant, sub, dlduv[idx].cpu(), dldp[idx].cpu(), dtype=torch.float32)
    rep = parity_report(mine, stack3(*t64[:3]), stack3(*y32[:3]))
    assert rep["frac_bad"] <= 0.01, rep
    # permutation equivariance on the sub-sample (fresh launch on gathered records)
    import epsm_mitsuba3_amd as epsm
    fp, lg, dg = epsm.calc_grad(variant, _gpu(sub, dev), dlduv[idx], dldp[idx], clip=0.1)
    assert torch.equal(stack3(fp, lg, dg), mine)


def test_product_refuses_cpu_tensors():
    import epsm_mitsuba3_amd as epsm
    from epsm_mitsuba3_amd.synth import synth_path_info
    pi, dlduv, dldp = synth_path_info(8, 2, seed=0)
    with pytest.raises(epsm.EpsmError):
        epsm.calc_grad("manifold", pi, dlduv, dldp)
