"""Bias check of exp/human.py's gradient: pose gradient averaged over seeds at pose 0, at the target pose, halfway."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from epsm_mitsuba3_amd import load_dict
from epsm_mitsuba3_amd.optim import to_ldr, resize
from epsm_mitsuba3_amd.matcher import Matcher
from epsm_mitsuba3_amd.exp import human as tasks
dev = "cuda"
scene = tasks.load_scene(dev)
integ = load_dict({"type": "manifold", "max_depth": tasks.max_depth})
gt_spp = int(sys.argv[1]) if len(sys.argv) > 1 else 512
gt = tasks.gt_scene(dev).render_primal(sensor=0, seed=0, spp=gt_spp, max_depth=tasks.max_depth)
gt_low = resize(to_ldr(gt), tasks.match_res)
matcher = Matcher(tasks.match_res, dev)
opt, apply_t, backward, output = tasks.optim_settings(scene)
params = scene.param_grads()
tp = tasks.target_pose().to(dev)
rep = tasks.resolution // tasks.match_res
mname = "match_" + tasks.matcher
for name, frac in (("pose 0", 0.0), ("halfway", 0.5), ("target", 1.0)):
    G = []
    for seed in range(8):
        with torch.no_grad():
            opt["pose"].copy_(tp * frac)
        apply_t(scene, opt)
        img = integ.render(scene, sensor=1, seed=seed, spp=tasks.spp)
        params = scene.param_grads() if params.flat.numel() != scene.param_grads().flat.numel() else params
        params.zero_()
        low = resize(to_ldr(img[..., :3]), tasks.match_res)
        g = getattr(matcher, mname)(low.reshape(-1, 3), gt_low.reshape(-1, 3)).reshape(tasks.match_res, tasks.match_res, 5).repeat(rep, rep, 1)
        integ.render_backward(scene, params, g, sensor=1, seed=seed, spp=tasks.spp)
        backward(opt, params)
        G.append(opt["pose"].grad.detach().clone().reshape(-1))
    G = torch.stack(G)
    mean, std = G.mean(0), G.std(0)
    ideal = (tp * frac - tp).reshape(-1)               # gradient of 0.5 |pose - target|^2
    cos = float(torch.nn.functional.cosine_similarity(mean, ideal, dim=0)) if frac < 1 else float("nan")
    print(f"{name:8s} |mean grad| {float(mean.norm()):.4e}  mean per-seed |grad| {float(G.norm(dim=1).mean()):.4e}  |std| {float(std.norm()):.4e}  cos(mean, pose - target) {cos:.3f}")
    top = mean.abs().argsort(descending=True)[:8]
    print("   largest components (joint, axis, mean, std):", [(int(i) // 3, int(i) % 3, round(float(mean[i]), 4), round(float(std[i]), 4)) for i in top])
