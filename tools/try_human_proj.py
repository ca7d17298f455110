"""exp/human.py: does the drift of the loop come from the component of the first-hit term ALONG THE VIEW RAY?

The reference turns the matcher's image-space motion (gx, gy) of a pixel into the displacement of the hit point INSIDE ITS
TRIANGLE when the camera ray moves accordingly (epsm.py:250-272: forward-mode derivative of si.p = sum b_j p_j with the
scene fixed) and hands b_j times that displacement to the triangle's vertices (epsm.py:561-562, 791-792).  On a triangle
tilted against the view direction that displacement is the screen-parallel motion the matcher asked for PLUS a component
along the view ray, 1/cos(tilt) times as long, which no image of a uniformly coloured body restrains.

    python tools/try_human_proj.py ITER LR [MODE...]     MODE: ref | proj | nofirst | noshadow

ref:      the reference's formula (what the library computes)
proj:     the first-hit displacement of paths that hit the BODY projected onto the plane perpendicular to the view ray
          (an experiment, NOT the reference's formula)
nofirst:  first-hit term of body hits dropped (only the occluder / shadow term moves the body)
noshadow: occluder term dropped (shadow record not used)
Prints, per mode, the history of the mean vertex distance to the target, the image MSE and the pose error."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import epsm_mitsuba3_amd as epsm
from epsm_mitsuba3_amd import integrators, optim
from epsm_mitsuba3_amd.exp import human
from epsm_mitsuba3_amd.records import PackedRecords, PackedScatter
from epsm_mitsuba3_amd.tangent_scatter import first_vertex_tangent, manifold_grad_scatter

its = int(sys.argv[1]) if len(sys.argv) > 1 else 60
lr = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
modes = sys.argv[3:] or ["ref", "proj"]
for k in ("matcher", "spp"):
    if os.environ.get("HUMAN_" + k.upper()):
        setattr(human, k, type(getattr(human, k))(os.environ["HUMAN_" + k.upper()]))


class Probe(integrators.ManifoldIntegrator):
    mode = "ref"
    body = (0, 0)

    def backward_from_trace(self, trace, params, grad_in, packed=None, out=None, mark=None, fused=None):
        dev = trace.ray_d.device
        rec, sc = PackedRecords(trace.path_info, device=dev), PackedScatter(trace.scatter_info, device=dev)
        first = trace.path_info[1]
        dlduv, dldp, grad_o = first_vertex_tangent(trace.ray_o, trace.ray_d, trace.ray_dx, trace.ray_dy, grad_in, trace.spp, trace.res,
                                                   first["points"][0], first["points"][1], first["points"][2], first["active"],
                                                   dlduv_width=2, want_origin_grad=True, path_offset=trace.path_offset)
        tri = trace.scatter_info[0]["tri"].long()
        on_body = (tri >= self.body[0]) & (tri < self.body[1])
        if self.mode == "proj":
            d = trace.ray_d / trace.ray_d.norm(dim=1, keepdim=True)
            perp = dldp - d * (d * dldp).sum(dim=1, keepdim=True)
            dldp = torch.where(on_body[:, None], perp, dldp)
        elif self.mode == "nofirst":
            # the body's own first-hit rows vanish, its shadow (paths that hit the floor first) still moves it
            dldp = torch.where(on_body[:, None], torch.zeros_like(dldp), dldp)
        if self.mode == "noshadow" and sc.packed[0].get("shadow") is not None:
            sc.packed[0]["shadow"][:, 0] = -1              # no occluder triangle
        manifold_grad_scatter(self.variant, rec, sc, dlduv, dldp.contiguous(), params.pos, params.nrm,
                              params.alpha if params.B else None, clip=self.outlier_clip)
        params.cam_origin += grad_o


for mode in modes:
    extra = []
    orig = human.optim_settings

    def wrapped(scene, orig=orig):
        opt, a, b, out = orig(scene)
        gt = human.gt_scene(scene.device).render_primal(sensor=0, seed=777, spp=256, max_depth=human.max_depth)

        def out2(o):
            e = out(o)
            a(scene, o)
            img = scene.render_primal(sensor=0, seed=778, spp=256, max_depth=human.max_depth)
            extra.append(float(((img[..., :3] - gt[..., :3]) ** 2).mean()))
            return e
        Probe.body = scene.mesh_tri_slices["human"]
        return opt, a, b, out2
    human.optim_settings = wrapped
    Probe.mode = mode
    integrators.register_integrator("manifold", lambda props: Probe({**props, "packed_log": False}))
    hist, opt = optim.run("manifold", "human", iterations=its, lr=lr, log=lambda s: None)
    human.optim_settings = orig
    tp = human.target_pose().reshape(-1)
    p = opt["pose"].detach().cpu().reshape(-1)
    print(f"## mode {mode}: {its} iterations, lr {lr}, matcher {human.matcher}")
    print("history", [round(h, 4) for h in hist])
    print("image mse x1e4", [round(e * 1e4, 2) for e in extra])
    print("pose error", round(float((p - tp).norm()), 4), "of", round(float(tp.norm()), 4),
          " last-20 mean distance", round(sum(hist[-20:]) / 20, 4), " best", round(min(hist), 4), flush=True)
