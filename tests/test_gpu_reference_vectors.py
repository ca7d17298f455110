"""The reference-held known answers of tests/golden/reference_vectors.py against the DEVICE code, through the C ABI of the
product library: epsm_probe (the tracer's TEA / PCG32 / microfacet / Fresnel / filter / camera functions as the kernels run
them), epsm_first_vertex_tangent, epsm_scatter and the one-launch epsm_backward_pass.  Same assertions as the CPU half
(tests/_refvec.py)."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests import _refvec as R
from tests.golden import reference_vectors as RV

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda", 0)


def device_probe(dev):
    from epsm_mitsuba3_amd import _lib

    def probe(what, rows, cfg=None):
        rows = torch.from_numpy(np.ascontiguousarray(rows, dtype=np.float32).view(np.int32)).to(dev)     # bits, not values
        out = torch.zeros((rows.shape[0], R.PROBE_OUT), dtype=torch.int32, device=dev)
        lib = _lib.lib()
        lib.epsm_probe.restype = C.c_int
        with torch.cuda.device(dev):
            rc = lib.epsm_probe(C.c_int(what), C.c_int64(rows.shape[0]), C.c_void_p(rows.data_ptr()), C.c_void_p(out.data_ptr()),
                                C.byref(cfg) if cfg is not None else None, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        _lib.check(rc, "epsm_probe")
        torch.cuda.synchronize()
        return out.cpu().numpy().view(np.float32)
    return probe


@pytest.mark.parametrize("check", R.ALL_PROBE_CHECKS, ids=lambda f: f.__name__)
def test_tracer_functions_on_the_device(check, dev):
    check(device_probe(dev))


def test_sampler_stream_on_the_device(dev):
    from epsm_mitsuba3_amd.integrators import sample_tea_32
    R.check_sampler_stream(device_probe(dev), sample_tea_32)


def test_probe_refuses_bad_arguments(dev):
    from epsm_mitsuba3_amd import _lib
    lib = _lib.lib()
    lib.epsm_probe.restype = C.c_int
    x = torch.zeros((1, 16), device=dev)
    assert lib.epsm_probe(C.c_int(99), C.c_int64(1), C.c_void_p(x.data_ptr()), C.c_void_p(x.data_ptr()), None, None) == -22
    assert lib.epsm_probe(C.c_int(R.PROBE["PRIMARY_RAY"]), C.c_int64(1), C.c_void_p(x.data_ptr()), C.c_void_p(x.data_ptr()), None, None) == -22


def test_first_vertex_tangent_kernel_matches_test_mesh(dev):
    from epsm_mitsuba3_amd.tangent_scatter import first_vertex_tangent

    def tangent(o, d, dx, dy, gx, gy, p0, p1, p2):
        t = lambda a: torch.tensor(np.asarray(a, dtype=np.float64).reshape(1, 3), dtype=torch.float32, device=dev)
        g = torch.zeros((1, 1, 5), device=dev); g[0, 0, 3], g[0, 0, 4] = gx, gy
        uv, dp, _ = first_vertex_tangent(t(o), t(d), t(dx), t(dy), g, 1, 1, t(p0), t(p1), t(p2), torch.ones(1, dtype=torch.bool, device=dev))
        return float(uv[0, 0, 0]), float(uv[0, 0, 1]), dp[0].cpu().double().numpy()
    R.check_tangent(tangent)


def test_scatter_kernel_matches_test_mesh(dev):
    from epsm_mitsuba3_amd.records import PackedRecords, PackedScatter
    from epsm_mitsuba3_amd.synth import path_info_to
    from epsm_mitsuba3_amd.tangent_scatter import scatter

    def run(pi, si, op, ol, od, V):
        rec = PackedRecords(path_info_to(pi, device=dev), device=dev)
        sc = PackedScatter(si, device=dev)
        gp = torch.zeros((V, 3), device=dev); gn = torch.zeros((V, 3), device=dev)
        scatter("manifold", rec, sc, op.to(dev), ol.to(dev), od.to(dev), gp, gn, None)
        torch.cuda.synchronize()
        assert float(gn.abs().max()) == 0.0
        return gp.cpu().double().numpy()
    R.check_scatter(run)


def test_backward_pass_emits_the_flat_normal_rows_of_test_mesh(dev):
    """The ONE-LAUNCH kernel (epsm_backward_cp.hip: tangent + calc_grad + scatter) on the reference's rectangle hit: whatever
    g_n calc_grad produces for the vertex, the rows it leaves in vertex_positions through the flat-normal branch must be g_n
    contracted with the Jacobian test_mesh.py:611-639 states (rows n.x, n.y; n.z has no first-order dependence), plus the
    direct p0/p1/p2 terms and the barycentric spread of diffuse_grad -- all of which the dense route (epsm_manifold_grad +
    the rows above) gives independently."""
    import epsm_mitsuba3_amd as epsm
    from epsm_mitsuba3_amd.records import PackedRecords, PackedScatter
    from epsm_mitsuba3_amd.synth import path_info_to
    from epsm_mitsuba3_amd.manifold_grad import manifold_grad_packed
    from epsm_mitsuba3_amd.tangent_scatter import manifold_grad_scatter
    pi, si, (b0, b1, b2) = R.rect_path_info()
    rec = PackedRecords(path_info_to(pi, device=dev), device=dev)
    sc = PackedScatter(si, device=dev)
    dlduv = torch.tensor([[2e-3, -1e-3]], device=dev); dldp = torch.tensor([[1e-3, 2e-3, -1e-3]], device=dev)
    fp, lg, dg = manifold_grad_packed("manifold", rec, dlduv, dldp, dlduv_cols=2)
    fp, dg = fp.cpu().double().numpy()[:, 0], dg.cpu().double().numpy()[:, 0]
    assert np.abs(fp[3]).max() > 0
    J = {name: np.asarray(row, dtype=np.float64).reshape(4, 3) for name, row in RV.SCATTER_ROWS}
    want = fp[3][0] * J["sh_frame.n.x"] + fp[3][1] * J["sh_frame.n.y"]
    for j, (v, b) in enumerate(zip(RV.RECT_UPPER, (b0, b1, b2))):
        want[v] += fp[j] + b * dg[0]
    gp = torch.zeros((4, 3), device=dev); gn = torch.zeros((4, 3), device=dev)
    manifold_grad_scatter("manifold", rec, sc, dlduv, dldp, gp, gn, None)
    torch.cuda.synchronize()
    got = gp.cpu().double().numpy()
    assert np.allclose(got, want, rtol=2e-4, atol=2e-5 * np.abs(want).max()), (got, want)
