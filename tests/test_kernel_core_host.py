"""The per-path kernel code (epsm_mitsuba3_amd/csrc/epsm_path_core.h) compiled for
the host CPU by tests/host_harness, against the reference goldens and the oracle.
This checks the block-adjoint algebra the gfx950 kernels run without a GPU; the
GPU tests (-m gpu) check the same code compiled by hipcc."""
import pytest
import torch

from _util import golden_files, golden_id, load_golden, stack3, parity_report
from host_core import host_core_calc_grad
from oracle.binding import oracle_calc_grad

FILES = golden_files()


@pytest.mark.parametrize("path", FILES, ids=golden_id)
def test_core_f64_matches_reference_f64(path):
    variant, pi, dlduv, dldp, ref = load_golden(path, dtype=torch.float64)
    fp, lg, dg = host_core_calc_grad(variant, pi, dlduv, dldp, dtype=torch.float64)
    mine = stack3(fp, lg, dg)
    truth = torch.cat([ref["ref64_param"], ref["ref64_light"], ref["ref64_diffuse"]]).double()
    assert not torch.isnan(mine).any()
    scale = truth.abs().amax(dim=(0, 2)).clamp_min(1e-12)
    rel = (mine - truth).abs().amax(dim=(0, 2)) / scale
    assert float(rel.max()) < 1e-6, float(rel.max())
    assert torch.equal(mine == 0, truth == 0)


@pytest.mark.parametrize("path", FILES, ids=golden_id)
def test_core_f32_matches_reference(path):
    variant, pi, dlduv, dldp, ref = load_golden(path, dtype=torch.float32)
    fp, lg, dg = host_core_calc_grad(variant, pi, dlduv, dldp, dtype=torch.float32)
    truth = torch.cat([ref["ref64_param"], ref["ref64_light"], ref["ref64_diffuse"]]).double()
    yard = torch.cat([ref["ref32_param"], ref["ref32_light"], ref["ref32_diffuse"]]).double()
    rep = parity_report(stack3(fp, lg, dg), truth, yard)
    assert rep["frac_bad"] <= 0.03, rep
    assert rep["median_rel"] < 1e-4, rep


@pytest.mark.parametrize("variant,profile", [("manifold", "bathroom"), ("manifold", "specular"),
                                              ("manifold_caustic", "pool"), ("manifold_caustic", "mixed")])
@pytest.mark.parametrize("K", [1, 3, 5])
def test_core_f64_matches_oracle_synthetic(variant, profile, K):
    from epsm_mitsuba3_amd.synth import synth_path_info
    pi, dlduv, dldp = synth_path_info(3000, K, seed=K, profile=profile, dtype=torch.float64, tangent_scale=1e-5)
    fp, lg, dg = host_core_calc_grad(variant, pi, dlduv, dldp, dtype=torch.float64)
    t = oracle_calc_grad(variant, pi, dlduv, dldp, dtype=torch.float64)
    mine, truth = stack3(fp, lg, dg), stack3(*t[:3])
    scale = truth.abs().amax(dim=(0, 2)).clamp_min(1e-12)
    rel = (mine - truth).abs().amax(dim=(0, 2)) / scale
    # a handful of nearly singular paths lose digits even in float64
    assert float(rel.quantile(0.999)) < 1e-7 and float(rel.max()) < 1e-3, (float(rel.max()))
