"""exp/human.py AT THE TARGET POSE: where does the non-zero mean gradient come from?  render_backward is linear in the
gradient image (up to the +-0.1 clamp of the per-path terms), so E[pose gradient] = A E[D] with D the matcher's image-space
displacement field: either the matcher's field has a non-zero MEAN between two renders of the same pose (different sample
counts, tone mapping, 8-bit quantisation), or the mapping A is not linear where it matters (the clamp).

    python tools/try_human_bias.py [SEEDS] [MATCHER] [RENDER_SPP] [GT_SPP]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from epsm_mitsuba3_amd import load_dict
from epsm_mitsuba3_amd.exp import human as tasks
from epsm_mitsuba3_amd.matcher import Matcher
from epsm_mitsuba3_amd.optim import resize, to_ldr

dev = "cuda"
S = int(sys.argv[1]) if len(sys.argv) > 1 else 12
mname = "match_" + (sys.argv[2] if len(sys.argv) > 2 else tasks.matcher)
spp = int(sys.argv[3]) if len(sys.argv) > 3 else tasks.spp
gt_spp = int(sys.argv[4]) if len(sys.argv) > 4 else 512
scene = tasks.load_scene(dev)
integ = load_dict({"type": "manifold", "max_depth": tasks.max_depth})
gt = tasks.gt_scene(dev).render_primal(sensor=0, seed=0, spp=gt_spp, max_depth=tasks.max_depth)
gt_low = resize(to_ldr(gt), tasks.match_res)
matcher = Matcher(tasks.match_res, dev)
opt, apply_t, backward, output = tasks.optim_settings(scene)
tp = tasks.target_pose().to(dev)
rep = tasks.resolution // tasks.match_res
R = tasks.match_res
with torch.no_grad():
    opt["pose"].copy_(tp)
apply_t(scene, opt)


def pose_grad(g_img, seed):
    apply_t(scene, opt)                                            # (a fresh autograd graph of the body model)
    params = scene.param_grads()
    integ.render_backward(scene, params, g_img.repeat(rep, rep, 1), sensor=1, seed=seed, spp=spp)
    backward(opt, params)
    return opt["pose"].grad.detach().clone().reshape(-1)


D, lows = [], []
for seed in range(S):
    img = integ.render(scene, sensor=1, seed=seed, spp=spp)
    low = resize(to_ldr(img[..., :3]), R)
    lows.append(low)
    D.append(getattr(matcher, mname)(low.reshape(-1, 3), gt_low.reshape(-1, 3)).reshape(R, R, 5))
D = torch.stack(D)                                              # (S,R,R,5)
M = D.mean(0)
rms = lambda t: float(t.pow(2).mean().sqrt())
print(f"# {mname}, render {spp} spp vs target {gt_spp} spp, {S} seeds, AT the target pose")
print(f"image: mean render - target (LDR, resized) {float((torch.stack(lows).mean(0) - gt_low).mean()):+.5f}, rms per-seed difference "
      f"{rms(torch.stack(lows) - gt_low[None]):.5f}")
print(f"displacement field (channels 3,4 = x,y of the 5-D points r,g,b,x,y: matcher.py:51-63): rms of one seed {rms(D[..., 3:5]):.4e}, rms of the MEAN over seeds {rms(M[..., 3:5]):.4e} "
      f"(pure noise would give {rms(D[..., 3:5]) / S ** 0.5:.4e})")
body = (gt_low - gt_low[0, 0]).abs().sum(-1) > 0.05               # crude: pixels that differ from the corner's floor colour
print(f"   of the mean field's energy, on pixels that differ from the floor colour ({float(body.float().mean()):.2f} of the image): "
      f"{float(M[..., 3:5][body].pow(2).sum() / M[..., 3:5].pow(2).sum()):.2f}")
g_full = torch.stack([pose_grad(D[s], s) for s in range(S)])
g_debiased = torch.stack([pose_grad(D[s] - M, s) for s in range(S)])
g_mean_only = torch.stack([pose_grad(M, s) for s in range(S)])
for name, g in (("matcher's field D_s", g_full), ("D_s - mean field", g_debiased), ("mean field alone", g_mean_only)):
    print(f"pose gradient from {name:20s}: |mean over seeds| {float(g.mean(0).norm()):.4e}, mean |per seed| {float(g.norm(dim=1).mean()):.4e}")
