"""Shared by the per-path tests of the accumulating backward kernel (test_gpu_backward_per_path.py at 20 000 paths,
test_gpu_full_size_packed.py on 4 096 paths cut out of a full-size slab): PRIVATE parameter rows per (path, vertex) and the
float64 oracle's lists mapped onto them.  Test infrastructure.

The kernel never writes calc_grad's per-path lists -- it adds rows into the parameter buffers -- so every (path, vertex)
gets its own hit triangle, its own emitter triangle and its own alpha slot.  The buffers then hold, row by row, the
per-path quantities the reference's replay would scatter (epsm.py:559-562, 622-627, 644-645):

    pos[hit(p,k), j]     = final_param_grad[5(k-1)+j][p] + b_j * diffuse_grad[k-1][p]
    nrm[hit(p,k), j]     = b_j * (g - sh (sh.g)) / |n|,  g = final_param_grad[5(k-1)+3][p]      (mesh.cpp:784-790)
    pos[emitter(p,k), j] = eb_j * eweight * light_grad[k-1][p]
    alpha[slot(p,k)]     = d hf / d alpha . final_param_grad[5(k-1)+4][p]
"""
import torch

from _util import gated_parity_report


def private_addressing(N, K, dev, gen):
    """Triangle table and per-vertex scatter records in which nothing is shared between (path, vertex) pairs."""
    NK = N * K
    t = torch.arange(NK, dtype=torch.int64)
    hit = torch.stack([3 * t, 3 * t + 1, 3 * t + 2, (4 | 8 | 1) | ((t + 1) << 8)], dim=1)          # attached, vertex normals, slot t
    te = NK + t
    emi = torch.stack([3 * te, 3 * te + 1, 3 * te + 2, torch.full_like(t, 4)], dim=1)
    table = torch.cat([hit, emi]).to(torch.int32)
    bits = lambda x: x.to(torch.float32).contiguous().view(torch.int32)
    si = []
    for k in range(1, K + 1):
        tk = (k - 1) * N + torch.arange(N, dtype=torch.int64)
        eb0 = torch.rand(N, generator=gen) * 0.5
        eb1 = torch.rand(N, generator=gen) * 0.5
        ew = 0.5 + torch.rand(N, generator=gen)
        dhf = torch.randn(N, 3, generator=gen)
        si.append({"tri": tk.to(torch.int32).to(dev),
                   "aux": torch.cat([tk.to(torch.int32)[:, None], bits(dhf)], dim=1).to(dev),
                   "emit": torch.stack([(NK + tk).to(torch.int32), bits(eb0), bits(eb1), bits(ew)], dim=1).to(dev),
                   "shadow": None, "table": table.to(dev),
                   "_f": (eb0.double(), eb1.double(), ew.double(), dhf.double())})
    return table.to(dev), si


def check_private_rows(kind, trace, si, params, grad_in, K, label=""):
    """``params`` (V = 6 N K rows, B = N K slots) after the backward pass on ``trace`` with the addressing ``si`` of
    ``private_addressing``: every row against the float64 oracle under the conditioning gate of SURVEY.md 8c."""
    from epsm_mitsuba3_amd.synth import path_info_to
    from oracle.binding import oracle_calc_grad, oracle_cond, oracle_first_vertex_tangent
    N, spp, res = int(trace.ray_d.shape[0]), trace.spp, trace.res
    off = int(trace.path_offset)
    # ---- the oracle, float64: calc_grad lists, then the linear map of the replay written out per path.  The tangents the
    # oracle is given are the stand-alone tangent kernel's (the same arithmetic, epsm_tangent_core.h, as inside the one-launch
    # kernel; against ITS oracle: test_gpu_tangent_scatter.py) -- a float32 Moeller-Trumbore derivative carries its own
    # conditioning, which is not what this test prices; the float64 tangent checks the camera-origin sum below.
    from epsm_mitsuba3_amd.tangent_scatter import first_vertex_tangent
    pi = path_info_to(trace.path_info, device="cpu")
    first = pi[1]
    _, _, go = oracle_first_vertex_tangent(trace.ray_o, trace.ray_d, trace.ray_dx, trace.ray_dy, grad_in, spp, res,
                                           first["points"][0], first["points"][1], first["points"][2], first["active"], 2, off)
    f1 = trace.path_info[1]
    dlduv, dldp, _ = first_vertex_tangent(trace.ray_o, trace.ray_d, trace.ray_dx, trace.ray_dy, grad_in, spp, res,
                                          f1["points"][0], f1["points"][1], f1["points"][2], f1["active"], dlduv_width=2, path_offset=off)
    dlduv, dldp = dlduv.double().cpu(), dldp.double().cpu()
    fp, lg, dg, _ = oracle_calc_grad(kind, pi, dlduv, dldp, dtype=torch.float64)
    cond = oracle_cond(kind, pi, dlduv, dldp)
    pos = params.pos.double().cpu().view(2, K, N, 3, 3)      # [hit / emitter][k][path][j][xyz]
    nrm = params.nrm.double().cpu().view(2, K, N, 3, 3)
    alpha = params.alpha.double().cpu().view(K, N)
    mine, truth = [], []
    zero = torch.zeros((N, 3), dtype=torch.float64)
    for k in range(1, K + 1):
        r = pi[k]
        b0, b1 = r["uv"][0].double(), r["uv"][1].double()
        bs = (b0, b1, 1.0 - b0 - b1)
        eb0, eb1, ew, dhf = si[k - 1]["_f"]
        ebs = (eb0, eb1, 1.0 - eb0 - eb1)
        has_nm = 5 * (k - 1) + 4 < len(fp)               # manifold_caustic: the last vertex has no n, m entries
        n = sum(r["normals"][j].double() * bs[j][:, None] for j in range(3))
        il = 1.0 / n.norm(dim=1, keepdim=True)
        sh = n * il
        g = fp[5 * (k - 1) + 3].double() if has_nm else zero
        pg = (g - sh * (sh * g).sum(dim=1, keepdim=True)) * il
        pg = torch.where(g.abs().sum(dim=1, keepdim=True) > 0, pg, torch.zeros_like(pg))
        for j in range(3):
            gp = fp[5 * (k - 1) + j].double() if 5 * (k - 1) + j < len(fp) else zero
            mine.append(pos[0, k - 1, :, j]); truth.append(gp + dg[k - 1].double() * bs[j][:, None])
            mine.append(nrm[0, k - 1, :, j]); truth.append(pg * bs[j][:, None])
            mine.append(pos[1, k - 1, :, j]); truth.append(lg[k - 1].double() * (ebs[j] * ew)[:, None])
        a = (fp[5 * (k - 1) + 4].double() * dhf).sum(dim=1) if has_nm else torch.zeros(N, dtype=torch.float64)
        mine.append(torch.stack([alpha[k - 1], torch.zeros(N, dtype=torch.float64), torch.zeros(N, dtype=torch.float64)], dim=1))
        truth.append(torch.stack([a, torch.zeros_like(a), torch.zeros_like(a)], dim=1))
    mine, truth = torch.stack(mine), torch.stack(truth)
    assert not torch.isnan(mine).any()
    assert float(truth.abs().max()) > 0
    # components of the LISTS within 2 % of the clamp make their path's rows discontinuous: those paths are set aside
    lists = torch.stack([t.double() for t in list(fp) + list(lg) + list(dg)])
    near = ((lists.abs() > 0.098) & (lists.abs() < 0.102)).any(dim=2).any(dim=0)
    rep = gated_parity_report(mine[:, ~near], truth[:, ~near], cond[~near], clip=0.0)
    print(kind, label, K, rep, "paths near the clamp:", int(near.sum()))
    assert rep["gate_share"] > 0.9, rep
    assert rep["frac_bad_inside"] <= 0.005, rep
    assert rep["median_rel"] < 1e-4, rep
    # masked paths contribute exact zeros; the emitter half of the normal buffer is never touched
    dead = (truth == 0).all(dim=2).all(dim=0)
    assert bool((mine[:, dead] == 0).all())
    assert float(nrm[1].abs().max()) == 0.0
    # camera origin: minus the sum of the ray-direction tangents (epsm.py:260-261)
    assert torch.allclose(params.cam_origin.double().cpu(), go, rtol=2e-4, atol=2e-4 * float(go.abs().max()))
    return rep
