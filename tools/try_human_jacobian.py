"""exp/human.py: is the pseudo-gradient field RESTORING around the target pose?  J_ij = d g_i / d theta_j by central
differences of the seed-averaged pose gradient g (the field the optimiser follows, theta <- theta - lr g) around the target
pose.  For the gradient of a loss with its minimum there J is the Hessian: symmetric positive semi-definite.  A direction
v with v^T J v < 0 is one along which the field pushes AWAY from the target.

    python tools/try_human_jacobian.py [EPS] [SEEDS] [MODE]      MODE: ref | first | shadow   (which terms move the body)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from epsm_mitsuba3_amd import integrators, load_dict
from epsm_mitsuba3_amd.exp import human as tasks
from epsm_mitsuba3_amd.matcher import Matcher
from epsm_mitsuba3_amd.optim import resize, to_ldr
from epsm_mitsuba3_amd.records import PackedRecords, PackedScatter
from epsm_mitsuba3_amd.tangent_scatter import first_vertex_tangent, manifold_grad_scatter

dev = "cuda"
eps = float(sys.argv[1]) if len(sys.argv) > 1 else 0.03
S = int(sys.argv[2]) if len(sys.argv) > 2 else 3
mode = sys.argv[3] if len(sys.argv) > 3 else "ref"
mname = "match_" + os.environ.get("HUMAN_MATCHER", tasks.matcher)


class Probe(integrators.ManifoldIntegrator):
    body = (0, 0)

    def backward_from_trace(self, trace, params, grad_in, packed=None, out=None, mark=None, fused=None):
        d_ = trace.ray_d.device
        rec, sc = PackedRecords(trace.path_info, device=d_), PackedScatter(trace.scatter_info, device=d_)
        first = trace.path_info[1]
        dlduv, dldp, grad_o = first_vertex_tangent(trace.ray_o, trace.ray_d, trace.ray_dx, trace.ray_dy, grad_in, trace.spp, trace.res,
                                                   first["points"][0], first["points"][1], first["points"][2], first["active"],
                                                   dlduv_width=2, want_origin_grad=True, path_offset=trace.path_offset)
        tri = trace.scatter_info[0]["tri"].long()
        on_body = (tri >= self.body[0]) & (tri < self.body[1])
        if mode == "shadow":
            dldp = torch.where(on_body[:, None], torch.zeros_like(dldp), dldp)
        if mode == "first" and sc.packed[0].get("shadow") is not None:
            sc.packed[0]["shadow"][:, 0] = -1
        manifold_grad_scatter(self.variant, rec, sc, dlduv, dldp.contiguous(), params.pos, params.nrm,
                              params.alpha if params.B else None, clip=self.outlier_clip)


scene = tasks.load_scene(dev)
integ = load_dict({"type": "manifold", "max_depth": tasks.max_depth}) if mode == "ref" else Probe({"max_depth": tasks.max_depth, "packed_log": False})
gt = tasks.gt_scene(dev).render_primal(sensor=0, seed=0, spp=512, max_depth=tasks.max_depth)
gt_low = resize(to_ldr(gt), tasks.match_res)
matcher = Matcher(tasks.match_res, dev)
opt, apply_t, backward, output = tasks.optim_settings(scene)
Probe.body = scene.mesh_tri_slices["human"]
tp = tasks.target_pose().to(dev)
rep = tasks.resolution // tasks.match_res


def grad_at(pose):
    G = torch.zeros(72, device=dev)
    for seed in range(S):
        with torch.no_grad():
            opt["pose"].copy_(pose)
        apply_t(scene, opt)
        img = integ.render(scene, sensor=1, seed=seed, spp=tasks.spp)
        params = scene.param_grads()
        low = resize(to_ldr(img[..., :3]), tasks.match_res)
        g = getattr(matcher, mname)(low.reshape(-1, 3), gt_low.reshape(-1, 3)).reshape(tasks.match_res, tasks.match_res, 5).repeat(rep, rep, 1)
        integ.render_backward(scene, params, g, sensor=1, seed=seed, spp=tasks.spp)
        backward(opt, params)
        G += torch.nan_to_num(opt["pose"].grad.detach().reshape(-1))
    return G / S


J = torch.zeros(72, 72, device=dev)
for j in range(72):
    e = torch.zeros(1, 72, device=dev); e[0, j] = eps
    J[:, j] = (grad_at(tp + e) - grad_at(tp - e)) / (2 * eps)
Js = 0.5 * (J + J.T)
ev, V = torch.linalg.eigh(Js.double().cpu())
diag = torch.diag(J).cpu()
print(f"# {mname}, mode {mode}, eps {eps}, {S} seeds: J = d(pose gradient)/d(pose) at the target pose")
print(f"diagonal: {int((diag > 0).sum())} of 72 positive (restoring); mean {float(diag.mean()):+.3f}; the negative ones (joint, axis, value): "
      f"{[(int(i) // 3, int(i) % 3, round(float(diag[i]), 3)) for i in torch.nonzero(diag <= 0).reshape(-1)]}")
print(f"symmetric part: eigenvalues min {float(ev[0]):+.3f}, max {float(ev[-1]):+.3f}; {int((ev < 0).sum())} negative; sum of negative "
      f"{float(ev[ev < 0].sum()):+.3f} vs sum of positive {float(ev[ev > 0].sum()):+.3f}")
print(f"antisymmetric part: |J - J^T| / |J + J^T| = {float((J - J.T).norm() / (J + J.T).norm()):.3f}")
evc = torch.linalg.eigvals(J.double().cpu())
print(f"eigenvalues of J itself: min real part {float(evc.real.min()):+.3f}; {int((evc.real < 0).sum())} with negative real part")
for k in range(3):
    v = V[:, k]
    top = v.abs().argsort(descending=True)[:6]
    print(f"   eigenvector {k} (lambda {float(ev[k]):+.3f}): largest components (joint, axis, weight) "
          f"{[(int(i) // 3, int(i) % 3, round(float(v[i]), 2)) for i in top]}")
torch.save({"J": J.cpu(), "target": tp.cpu()}, os.environ.get("HUMAN_J_OUT", "/tmp/human_J.pt"))
