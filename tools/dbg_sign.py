import torch, importlib
import epsm_mitsuba3_amd as epsm
from epsm_mitsuba3_amd.optim import to_ldr, resize
from epsm_mitsuba3_amd.matcher import Matcher, sinkhorn_divergence
tasks = importlib.import_module("epsm_mitsuba3_amd.exp.plate")
dev="cuda"
scene = tasks.load_scene(dev)
integ = epsm.load_dict({"type":"manifold","max_depth":tasks.max_depth})
gt = tasks.gt_scene(dev).render_primal(sensor=0, seed=0, spp=512, max_depth=tasks.max_depth)
gt_low = resize(to_ldr(gt), tasks.match_res)
m = Matcher(tasks.match_res, dev)
opt, apply_t, backward, output = tasks.optim_settings(scene)
def loss_at(t):
    opt["trans"].data[:] = torch.tensor(t, device=dev)
    apply_t(scene, opt)
    img = integ.render(scene, sensor=1, seed=0, spp=64)
    low = resize(to_ldr(img[...,:3]), tasks.match_res)
    r = torch.cat([low.reshape(-1,3).clamp(0,1), m.pos],1); t_ = torch.cat([gt_low.reshape(-1,3).clamp(0,1), m.pos],1)
    return float(sinkhorn_divergence(r, t_)), low
for tx in (-0.2, 0.0, 0.2, 0.6):
    print("t_x", tx, "loss", loss_at([tx, 0.15*0, 0])[0])
# gradient at 0
l0, low = loss_at([0,0,0])
g = m.match_Sinkhorn(low.reshape(-1,3), gt_low.reshape(-1,3)).reshape(tasks.match_res, tasks.match_res, 5)
bright = low[...,0] > 0.6
print("bright pixels", int(bright.sum()), "mean gx, gy over bright", float(g[...,3][bright].mean()), float(g[...,4][bright].mean()))
ys, xs = torch.nonzero(bright, as_tuple=True); print("bright centroid (row,col)", float(ys.float().mean()), float(xs.float().mean()))
gb = gt_low[...,0] > 0.6; ys, xs = torch.nonzero(gb, as_tuple=True); print("target centroid (row,col)", float(ys.float().mean()), float(xs.float().mean()))
rep = tasks.resolution // tasks.match_res
params = scene.param_grads()
integ.render_backward(scene, params, g.repeat(rep, rep, 1), seed=0)
print("EPSM grad wrt translation:", params.mesh_pos("light").sum(0))
