"""exp/human.py, one step back from the pose: which way do the PER-VERTEX gradients push the body when the target is the
same body translated by 5 cm?  A rigid translation has an unambiguous answer -- every vertex should move along the
offset -- and splits the two terms: the first-hit term (body seen directly) and the occluder term (its shadow).

    python tools/try_human_translate.py [SEEDS]

Prints, per offset direction and per term, the mean of -grad over the body's vertices (the direction a gradient step moves
it), its cosine with the offset, and the share of the vertex gradients whose own step has a positive component along the offset."""
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
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
mname = "match_" + os.environ.get("HUMAN_MATCHER", tasks.matcher)


class Probe(integrators.ManifoldIntegrator):
    mode = "ref"
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
        if self.mode == "shadow only":
            dldp = torch.where(on_body[:, None], torch.zeros_like(dldp), dldp)
        if self.mode == "first hit only" and sc.packed[0].get("shadow") is not None:
            sc.packed[0]["shadow"][:, 0] = -1
        manifold_grad_scatter(self.variant, rec, sc, dlduv, dldp.contiguous(), params.pos, params.nrm,
                              params.alpha if params.B else None, clip=self.outlier_clip)


scene = tasks.load_scene(dev)
scene.attach("human", positions=True)
Probe.body = scene.mesh_tri_slices["human"]
integ = Probe({"max_depth": tasks.max_depth, "packed_log": False})
base = scene.vertex_positions("human").clone()
matcher = Matcher(tasks.match_res, dev)
rep = tasks.resolution // tasks.match_res
for name, off in (("+x (sideways in the image)", [0.05, 0, 0]), ("+y (away from the camera)", [0, 0.05, 0]), ("+z (up)", [0, 0, 0.05])):
    off_t = torch.tensor(off, device=dev)
    gt_scene = tasks.load_scene(dev)
    gt_scene.set_vertex_positions("human", base + off_t)
    gt = gt_scene.render_primal(sensor=0, seed=0, spp=512, max_depth=tasks.max_depth)
    gt_low = resize(to_ldr(gt), tasks.match_res)
    print(f"## target = the body translated by {off} m: {name}")
    for mode in ("ref", "first hit only", "shadow only"):
        Probe.mode = mode
        G = torch.zeros_like(base)
        for seed in range(seeds):
            img = integ.render(scene, sensor=1, seed=seed, spp=tasks.spp)
            params = scene.param_grads()
            low = resize(to_ldr(img[..., :3]), tasks.match_res)
            g = getattr(matcher, mname)(low.reshape(-1, 3), gt_low.reshape(-1, 3)).reshape(tasks.match_res, tasks.match_res, 5).repeat(rep, rep, 1)
            integ.render_backward(scene, params, g, sensor=1, seed=seed, spp=tasks.spp)
            G += torch.nan_to_num(params.mesh_pos("human"))
        step = -G / seeds                                     # what a gradient step does to the vertices
        mean = step.mean(dim=0)
        u = off_t / off_t.norm()
        cos = float((mean @ u) / mean.norm().clamp_min(1e-30))
        along = float(((step @ u) > 0).float().mean())
        moved = float((step.norm(dim=1) > 0).float().mean())
        print(f"   {mode:15s} mean step {[round(float(x), 5) for x in mean]}  cos with the offset {cos:+.3f}  "
              f"vertices with a gradient {moved:.2f}, of all vertices stepping along the offset {along:.2f}", flush=True)
