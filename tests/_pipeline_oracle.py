"""Float64 CPU restatement of one backward pass over PathTrace tiles (tangent ->
calc_grad -> scatter), assembled from the oracle pieces.  Test infrastructure."""
import torch

from epsm_mitsuba3_amd.synth import path_info_to
from oracle.binding import oracle_calc_grad, oracle_first_vertex_tangent, oracle_scatter


def oracle_backward(variant, traces, grad_in, V, B, clip=0.1, straddle_band=0.0):
    """``straddle_band`` > 0 additionally returns per-element allowances: how much each sum changes when the
    outlier threshold (epsm.py:932-944) moves by +-band, i.e. the weight of the per-path components so close
    to the threshold that fp32 and fp64 may land on different sides of it (reported separately)."""
    allow = [torch.zeros((V, 3), dtype=torch.float64), torch.zeros((V, 3), dtype=torch.float64),
             torch.zeros((B,), dtype=torch.float64)]
    gp = torch.zeros((V, 3), dtype=torch.float64)
    gn = torch.zeros((V, 3), dtype=torch.float64)
    ga = torch.zeros((B,), dtype=torch.float64)
    go = torch.zeros(3, dtype=torch.float64)
    for tr in traces:
        pi = path_info_to(tr.path_info, device="cpu")
        si = [{k: (v.cpu() if v is not None else None) for k, v in r.items()} for r in tr.scatter_info]
        first = pi[1]
        dlduv, dldp, o = oracle_first_vertex_tangent(
            tr.ray_o, tr.ray_d, tr.ray_dx, tr.ray_dy, grad_in, tr.spp, tr.res,
            first["points"][0], first["points"][1], first["points"][2], first["active"], 2, tr.path_offset)
        # calc_grad consumes fp32 tangents in the product; keep float64 here (the truth)
        fp, lg, dg, _ = oracle_calc_grad(variant, pi, dlduv, dldp, clip=clip, dtype=torch.float64)
        p, n, a = oracle_scatter(variant, pi, si, fp, lg, dg, V, B)
        gp += p; gn += n; ga += a; go += o
        if straddle_band > 0:
            band = []
            for c in (clip * (1 - straddle_band), clip * (1 + straddle_band)):
                fp2, lg2, dg2, _ = oracle_calc_grad(variant, pi, dlduv, dldp, clip=c, dtype=torch.float64)
                band.append(oracle_scatter(variant, pi, si, fp2, lg2, dg2, V, B))
            # ... and of the paths whose system is so ill-conditioned that the fp32 restatement itself leaves the
            # fp64 result (cond > 1e4 is outside the stated tolerance)
            fp3, lg3, dg3, _ = oracle_calc_grad(variant, pi, dlduv, dldp, clip=clip, dtype=torch.float32)
            f32 = oracle_scatter(variant, pi, si, [t.double() for t in fp3], [t.double() for t in lg3],
                                 [t.double() for t in dg3], V, B)
            for j, exact in enumerate((p, n, a)):
                allow[j] += (band[0][j] - band[1][j]).abs() + 2 * (f32[j].double() - exact).abs()
    if straddle_band > 0:
        return gp, gn, ga, go, allow
    return gp, gn, ga, go
