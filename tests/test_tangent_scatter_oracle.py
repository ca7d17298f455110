"""Known-answer tests that pin oracle/epsm_oracle_aux.c (the reference has no test for
these parts and they cannot be imported without Dr.Jit -- SURVEY.md 8c):

  * tangent: finite differences of a float64 ray/triangle intersection along grad_d;
  * scatter: torch autograd (float64) of the loss expressions the reference feeds to
    dr.backward (epsm.py:559-562, 626-627, 644-645) with the gathers and shading-normal
    formulas of mesh.h:94-106 / mesh.cpp:709,729,784-790,811-827 written out in torch.
"""
import numpy as np
import pytest
import torch

from epsm_mitsuba3_amd.synth import (synth_camera_rays, synth_first_hit_triangles, synth_path_info,
                                     synth_scatter_info)
from epsm_mitsuba3_amd.records import loose_from_packed
from oracle.binding import oracle_first_vertex_tangent, oracle_scatter, oracle_calc_grad


def _moeller_trumbore64(o, d, p0, p1, p2):
    e1, e2 = p1 - p0, p2 - p0
    pvec = np.cross(d, e2)
    inv = 1.0 / np.sum(e1 * pvec, -1)
    tvec = o - p0
    u = np.sum(tvec * pvec, -1) * inv
    qvec = np.cross(tvec, e1)
    v = np.sum(d * qvec, -1) * inv
    return u, v


def test_tangent_matches_finite_differences():
    res, spp = 8, 4
    o, d, dx, dy = synth_camera_rays(res, spp, seed=1)
    p0, p1, p2, b0, b1 = synth_first_hit_triangles(o, d, seed=1)
    g = torch.Generator().manual_seed(5)
    grad_in = torch.randn((12, 10, 5), generator=g)          # larger than res: the crop matters
    active = torch.ones(d.shape[0], dtype=torch.bool)
    active[::7] = False
    dlduv, dldp, go = oracle_first_vertex_tangent(o, d, dx, dy, grad_in, spp, res, p0, p1, p2, active)
    O, D, DX, DY, P0, P1, P2 = (t.double().numpy() for t in (o, d, dx, dy, p0, p1, p2))
    pix = np.arange(d.shape[0]) // spp
    gxy = grad_in.double().numpy()[pix // res, pix % res, 3:5]
    gd = (DX - D) * gxy[:, :1] + (DY - D) * gxy[:, 1:2]
    # the oracle's own (u,v) reproduce the logged barycentrics
    u0, v0 = _moeller_trumbore64(O, D, P0, P1, P2)
    assert np.allclose(u0, b1.double().numpy(), atol=1e-5) and np.allclose(1 - u0 - v0, b0.double().numpy(), atol=1e-5)
    h = 1e-6
    up, vp = _moeller_trumbore64(O, D + h * gd, P0, P1, P2)
    um, vm = _moeller_trumbore64(O, D - h * gd, P0, P1, P2)
    du, dv = (up - um) / (2 * h), (vp - vm) / (2 * h)
    act = active.numpy()
    assert np.allclose(dlduv[:, 0, 1].numpy()[act], du[act], rtol=1e-6, atol=1e-9)
    assert np.allclose(dlduv[:, 0, 0].numpy()[act], (-du - dv)[act], rtol=1e-6, atol=1e-9)
    dp = (P1 - P0) * du[:, None] + (P2 - P0) * dv[:, None]
    assert np.allclose(dldp.numpy()[act], dp[act], rtol=1e-6, atol=1e-9)
    assert np.all(dlduv.numpy()[~act] == 0) and np.all(dldp.numpy()[~act] == 0)
    assert np.allclose(go.numpy(), -gd.sum(0), rtol=1e-9)


def _autograd_scatter(variant, pi, si, fp, lg, dg, V, B):
    """The reference's dr.backward calls, written in torch float64."""
    K = len(pi) - 1
    N = pi[0]["cam"].shape[0]
    P = len(fp)
    pos_attached = torch.randn((V, 3), dtype=torch.float64).requires_grad_(True)
    nrm_attached = torch.randn((V, 3), dtype=torch.float64).requires_grad_(True)
    alpha = torch.zeros(max(B, 1), dtype=torch.float64, requires_grad=True)
    loss = torch.zeros((), dtype=torch.float64)
    for it in range(K):
        r, s = pi[it + 1], loose_from_packed(si[it])
        mode = s["mode"].long()
        vidx = s["vidx"].long()
        ok = ((vidx >= 0) & (vidx < V)).all(-1)
        b0, b1 = r["uv"][0].double(), r["uv"][1].double()
        bw = torch.stack([b0, b1, 1 - b0 - b1], -1)
        pts = [r["points"][j].double() for j in range(3)]            # logged (primal) positions
        has_nm = it * 5 + 4 < P
        for n in range(N):
            m = int(mode[n])
            if ok[n] and (m & 4):
                # attached gathers: value = logged position, gradient flows to the buffer row
                pj = [pts[j][n].detach() + (pos_attached[vidx[n, j]] - pos_attached[vidx[n, j]].detach()) for j in range(3)]
                if has_nm:                                          # epsm.py:559-560
                    loss = loss + sum((pj[j] * fp[5 * it + j][n].double()).sum() for j in range(3))
                follow_p = sum(pj[j] * bw[n, j] for j in range(3))   # mesh.cpp:709, FollowShape: b detached
                loss = loss + (follow_p * dg[it][n].double()).sum()  # epsm.py:561-562
            if has_nm:
                gn = fp[5 * it + 3][n].double()
                sgn = -1.0 if (m & 2) else 1.0
                if m & 1:
                    if ok[n] and (m & 8):
                        # buffer normal = sgn * logged normal (mesh.cpp:820-827 flips after the gather)
                        nb = [sgn * r["normals"][j][n].double().detach()
                              + (nrm_attached[vidx[n, j]] - nrm_attached[vidx[n, j]].detach()) for j in range(3)]
                        nn = sum(nb[j] * bw[n, j] for j in range(3))
                        sh = sgn * nn / nn.norm()                    # mesh.cpp:784-790, 820-827
                        loss = loss + (sh * gn).sum()                # epsm.py:645
                elif ok[n] and (m & 4):
                    pj = [pts[j][n].detach() + (pos_attached[vidx[n, j]] - pos_attached[vidx[n, j]].detach()) for j in range(3)]
                    c = torch.linalg.cross(pj[1] - pj[0], pj[2] - pj[0])
                    sh = sgn * c / c.norm()                          # mesh.cpp:729, 811, 820-827
                    loss = loss + (sh * gn).sum()
                bid = int(s["bsdf_id"][n])
                if 0 <= bid < B:
                    hf = s["dhf_dalpha"][n].double() * alpha[bid]    # hf depends linearly on alpha with this slope
                    loss = loss + (hf * fp[5 * it + 4][n].double()).sum()
            ev = s["evidx"][n].long()
            if ((ev >= 0) & (ev < V)).all() and (int(s["emode"][n]) & 4):     # si_direct.p attached with the emitter mesh
                c0, c1 = s["eb0"][n].double(), s["eb1"][n].double()
                ep = [pos_attached[ev[j]] for j in range(3)]
                direct_p = ep[0] * c0 + ep[1] * c1 + ep[2] * (1 - c0 - c1)
                loss = loss + (direct_p * (lg[it][n].double() * s["eweight"][n].double())).sum()   # epsm.py:626-627
            if it == 0 and "svidx" in s:                              # epsm.py:609-620 (max_depth <= 3)
                sv = s["svidx"][n].long()
                if ((sv >= 0) & (sv < V)).all() and (int(s["smode"][n]) & 4):
                    c0, c1 = s["sb0"][n].double(), s["sb1"][n].double()
                    sp = [pos_attached[sv[j]] for j in range(3)]
                    occluder_p = sp[0] * c0 + sp[1] * c1 + sp[2] * (1 - c0 - c1)                  # FollowShape: b detached
                    loss = loss + (occluder_p * dg[0][n].double() * s["sdis"][n].double()).sum()   # :617
    gp, gn_, ga = torch.autograd.grad(loss, [pos_attached, nrm_attached, alpha], allow_unused=True)
    z = lambda t, like: torch.zeros_like(like) if t is None else t
    return z(gp, pos_attached), z(gn_, nrm_attached), z(ga, alpha)[:B]


@pytest.mark.parametrize("variant", ["manifold", "manifold_caustic"])
def test_scatter_matches_autograd_of_reference_losses(variant):
    N, K, V, B = 160, 3, 40, 3
    pi, dlduv, dldp = synth_path_info(N, K, seed=3, profile="mixed", tangent_scale=1e-4)
    si = synth_scatter_info(N, K, V, seed=3, n_bsdfs=B, shadow=True)
    fp, lg, dg, _ = oracle_calc_grad(variant, pi, dlduv, dldp, dtype=torch.float64)
    # dense random "gradients" exercise every branch, not only the unmasked paths
    g = torch.Generator().manual_seed(0)
    fp = [x + 0.01 * torch.randn(x.shape, generator=g, dtype=torch.float64) for x in fp]
    lg = [x + 0.01 * torch.randn(x.shape, generator=g, dtype=torch.float64) for x in lg]
    dg = [x + 0.01 * torch.randn(x.shape, generator=g, dtype=torch.float64) for x in dg]
    gp, gn, ga = oracle_scatter(variant, pi, si, fp, lg, dg, V, B)
    rp, rn, ra = _autograd_scatter(variant, pi, si, fp, lg, dg, V, B)
    assert float(gp.abs().max()) > 0 and float(gn.abs().max()) > 0 and float(ga.abs().max()) > 0
    assert torch.allclose(gp, rp, rtol=1e-6, atol=1e-9), float((gp - rp).abs().max())
    assert torch.allclose(gn, rn, rtol=1e-6, atol=1e-9), float((gn - rn).abs().max())
    assert torch.allclose(ga, ra, rtol=1e-6, atol=1e-9)
