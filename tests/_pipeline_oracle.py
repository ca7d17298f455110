"""Float64 CPU restatement of one backward pass over PathTrace tiles (tangent ->
calc_grad -> scatter), assembled from the oracle pieces.  Test infrastructure."""
import torch

from epsm_mitsuba3_amd.synth import path_info_to
from oracle.binding import oracle_calc_grad, oracle_first_vertex_tangent, oracle_scatter


def oracle_backward(variant, traces, grad_in, V, B, clip=0.1):
    gp = torch.zeros((V, 3), dtype=torch.float64)
    gn = torch.zeros((V, 3), dtype=torch.float64)
    ga = torch.zeros((B,), dtype=torch.float64)
    go = torch.zeros(3, dtype=torch.float64)
    for tr in traces:
        pi = path_info_to(tr.path_info, device="cpu")
        si = [{k: v.cpu() for k, v in r.items()} for r in tr.scatter_info]
        first = pi[1]
        dlduv, dldp, o = oracle_first_vertex_tangent(
            tr.ray_o, tr.ray_d, tr.ray_dx, tr.ray_dy, grad_in, tr.spp, tr.res,
            first["points"][0], first["points"][1], first["points"][2], first["active"], 2, tr.path_offset)
        # calc_grad consumes fp32 tangents in the product; keep float64 here (the truth)
        fp, lg, dg, _ = oracle_calc_grad(variant, pi, dlduv, dldp, clip=clip, dtype=torch.float64)
        p, n, a = oracle_scatter(variant, pi, si, fp, lg, dg, V, B)
        gp += p; gn += n; ga += a; go += o
    return gp, gn, ga, go
