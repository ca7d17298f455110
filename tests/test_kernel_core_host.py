"""The kernels' arithmetic compiled for the host CPU by tests/host_harness, against the reference goldens and the
oracle: epsm_mitsuba3_amd/csrc/epsm_path_core.h (core "path": one lane per path, the dense calc_grad kernel) and
epsm_cp_core.h (core "cp": one lane per (path, constraint vertex), the fused backward kernel).  This checks the
block-adjoint algebra the gfx950 kernels run without a GPU; the GPU tests (-m gpu) check the same code compiled by
hipcc."""
import pytest
import torch

from _util import golden_files, golden_id, load_golden, stack3, parity_report, gated_parity_report
from host_core import host_core_calc_grad
from oracle.binding import oracle_calc_grad, oracle_cond

FILES = golden_files()
CORES = ["path", "cp"]


@pytest.mark.parametrize("core", CORES)
@pytest.mark.parametrize("path", FILES, ids=golden_id)
def test_core_f64_matches_reference_f64(path, core):
    variant, pi, dlduv, dldp, ref = load_golden(path, dtype=torch.float64)
    fp, lg, dg = host_core_calc_grad(variant, pi, dlduv, dldp, dtype=torch.float64, core=core)
    mine = stack3(fp, lg, dg)
    truth = torch.cat([ref["ref64_param"], ref["ref64_light"], ref["ref64_diffuse"]]).double()
    assert not torch.isnan(mine).any()
    scale = truth.abs().amax(dim=(0, 2)).clamp_min(1e-12)
    rel = (mine - truth).abs().amax(dim=(0, 2)) / scale
    assert float(rel.max()) < 1e-6, float(rel.max())
    assert torch.equal(mine == 0, truth == 0)


@pytest.mark.parametrize("core", CORES)
@pytest.mark.parametrize("path", FILES, ids=golden_id)
def test_core_f32_matches_reference(path, core):
    variant, pi, dlduv, dldp, ref = load_golden(path, dtype=torch.float32)
    fp, lg, dg = host_core_calc_grad(variant, pi, dlduv, dldp, dtype=torch.float32, core=core)
    truth = torch.cat([ref["ref64_param"], ref["ref64_light"], ref["ref64_diffuse"]]).double()
    yard = torch.cat([ref["ref32_param"], ref["ref32_light"], ref["ref32_diffuse"]]).double()
    rep = parity_report(stack3(fp, lg, dg), truth, yard)
    assert rep["frac_bad"] <= 0.01, rep
    assert rep["median_rel"] < 1e-4, rep
    # ... and without the yardstick: inside the conditioning gate of SURVEY.md 8c (cond_2 < 1e4, from the oracle) at most
    # 0.5 % of the paths leave scale * max(2e-4, 4 eps cond)
    cond = oracle_cond(variant, pi, dlduv, dldp)
    rep = gated_parity_report(stack3(fp, lg, dg), truth, cond)
    assert rep["n_bad_inside"] <= max(1, int(0.005 * truth.shape[1])), rep


@pytest.mark.parametrize("path", FILES, ids=golden_id)
def test_oracle_cond_against_the_reference_matrices(path):
    """``oracle_cond`` (max cond_2 over the solves a path USES, power iteration in C) against ``ref64_cond`` of the
    fixture (max over ALL matrices the reference's float64 run inverted, torch.linalg.cond): never larger, and equal
    wherever a path's deepest system is among the used ones (most paths)."""
    import numpy as np
    variant, pi, dlduv, dldp, ref = load_golden(path, dtype=torch.float64)
    mine = oracle_cond(variant, pi, dlduv, dldp)
    theirs = ref["ref64_cond"].double()
    finite = torch.isfinite(theirs)
    assert bool((mine[finite] <= theirs[finite] * (1 + 1e-6) + 1e-9).all())
    same = (mine[finite] - theirs[finite]).abs() <= 1e-3 * theirs[finite]
    # equal wherever the path's worst system is one it uses: nearly always on chains without a diffuse vertex
    assert float(same.double().mean()) > (0.85 if "specular" in path else 0.1), float(same.double().mean())


@pytest.mark.parametrize("variant,profile", [("manifold", "bathroom"), ("manifold", "specular"),
                                              ("manifold_caustic", "pool"), ("manifold_caustic", "mixed")])
@pytest.mark.parametrize("K", [1, 3, 5])
@pytest.mark.parametrize("core", CORES)
def test_core_f64_matches_oracle_synthetic(variant, profile, K, core):
    from epsm_mitsuba3_amd.synth import synth_path_info
    pi, dlduv, dldp = synth_path_info(3000, K, seed=K, profile=profile, dtype=torch.float64, tangent_scale=1e-5)
    fp, lg, dg = host_core_calc_grad(variant, pi, dlduv, dldp, dtype=torch.float64, core=core)
    t = oracle_calc_grad(variant, pi, dlduv, dldp, dtype=torch.float64)
    mine, truth = stack3(fp, lg, dg), stack3(*t[:3])
    scale = truth.abs().amax(dim=(0, 2)).clamp_min(1e-12)
    rel = (mine - truth).abs().amax(dim=(0, 2)) / scale
    # a handful of nearly singular paths lose digits even in float64
    assert float(rel.quantile(0.999)) < 1e-7 and float(rel.max()) < 1e-3, (float(rel.max()))


@pytest.mark.parametrize("variant,profile", [("manifold", "bathroom"), ("manifold", "specular"), ("manifold", "mixed"),
                                              ("manifold_caustic", "pool"), ("manifold_caustic", "caustic")])
@pytest.mark.parametrize("core", CORES)
def test_core_f32_inside_the_conditioning_gate(variant, profile, core):
    """N = 20 000, K = 5: inside cond_2 < 1e4 at most 0.5 % of the paths may leave
    scale * max(2e-4, 4 eps cond); the fraction outside the gate is reported, not bounded."""
    from epsm_mitsuba3_amd.synth import synth_path_info
    pi, dlduv, dldp = synth_path_info(20000, 5, seed=45, profile=profile, tangent_scale=2e-5)
    fp, lg, dg = host_core_calc_grad(variant, pi, dlduv, dldp, dtype=torch.float32, core=core)
    t = oracle_calc_grad(variant, pi, dlduv, dldp, dtype=torch.float64)
    cond = oracle_cond(variant, pi, dlduv, dldp)
    rep = gated_parity_report(stack3(fp, lg, dg), stack3(*t[:3]), cond)
    print(variant, profile, rep)
    assert rep["gate_share"] > 0.95, rep
    assert rep["frac_bad_inside"] <= 0.005, rep
