"""The CPU oracle (oracle/epsm_oracle.c) against golden vectors produced by the
reference's own calc_grad (tests/golden/gen_golden.py; epsm.py:745-1200)."""
import numpy as np
import pytest
import torch

from _util import golden_files, golden_id, load_golden, stack3, parity_report
from oracle.binding import oracle_calc_grad

FILES = golden_files()


def test_goldens_present():
    assert len(FILES) >= 16


@pytest.mark.parametrize("path", FILES, ids=golden_id)
def test_oracle_f64_matches_reference_f64(path):
    """Pins every mask / overwrite rule: in float64 rounding is ~1e-13, so any
    logic difference shows.  Exact-zero pattern must agree as well."""
    variant, pi, dlduv, dldp, ref = load_golden(path, dtype=torch.float64)
    fp, lg, dg, _ = oracle_calc_grad(variant, pi, dlduv, dldp, clip=0.1, dtype=torch.float64)
    assert len(fp) == ref["ref64_param"].shape[0]
    mine = stack3(fp, lg, dg)
    truth = torch.cat([ref["ref64_param"], ref["ref64_light"], ref["ref64_diffuse"]]).double()
    assert mine.shape == truth.shape
    scale = truth.abs().amax(dim=(0, 2)).clamp_min(1e-12)
    relerr = ((mine - truth).abs().amax(dim=(0, 2)) / scale)
    assert float(relerr.max()) < 1e-7, f"max rel err {float(relerr.max()):.3e}"
    assert torch.equal(mine == 0, truth == 0), "zero pattern differs from the reference"


@pytest.mark.parametrize("path", FILES, ids=golden_id)
def test_oracle_f32_matches_reference_f32(path):
    """fp32 oracle vs the reference as shipped (fp32 torch, LAPACK LU): both carry
    fp32 rounding, so the yardstick is the reference's own distance to float64."""
    variant, pi, dlduv, dldp, ref = load_golden(path, dtype=torch.float32)
    fp, lg, dg, _ = oracle_calc_grad(variant, pi, dlduv, dldp, clip=0.1, dtype=torch.float32)
    mine = stack3(fp, lg, dg)
    truth = torch.cat([ref["ref64_param"], ref["ref64_light"], ref["ref64_diffuse"]]).double()
    yard = torch.cat([ref["ref32_param"], ref["ref32_light"], ref["ref32_diffuse"]]).double()
    rep = parity_report(mine, truth, yard)
    assert rep["frac_bad"] <= 0.02, rep
    assert rep["median_rel"] < 1e-4, rep


def test_oracle_threads_and_clip_switch():
    variant, pi, dlduv, dldp, ref = load_golden(FILES[0], dtype=torch.float64)
    a = oracle_calc_grad(variant, pi, dlduv, dldp, clip=0.1, dtype=torch.float64, nthreads=1)
    b = oracle_calc_grad(variant, pi, dlduv, dldp, clip=0.1, dtype=torch.float64, nthreads=4)
    assert a[3] == 1 and b[3] == 4
    assert torch.equal(stack3(*a[:3]), stack3(*b[:3]))
    c = oracle_calc_grad(variant, pi, dlduv * 1e4, dldp, clip=0.0, dtype=torch.float64)
    assert float(stack3(*c[:3]).abs().max()) > 0.1  # clamp disabled
