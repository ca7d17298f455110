"""exp/human.py: is the drift of the loop the response to the FROZEN target image?  The reference matches every render
against ONE 512-spp render of the target converted to 8 bits (EPSM/optim_human.py:40-50); tools/try_human_bias.py shows that
at the target pose the matcher's displacement field then has a non-zero mean over render seeds, and that this mean field alone
accounts for the seed-mean of the pose gradient.  Here the loop of optim.run with

    fresh:    the target re-rendered under a new seed in every iteration (its noise no longer frozen)
    noquant:  tone mapping without the 8-bit rounding, both images
    clean:    the target rendered at 8192 spp (frozen, but nearly noise-free), no rounding

    prb:      NOT the manifold integrator: prb_reparam (true gradients of the L2 image loss, EPSM/optim_human.py:112-114) on
              sensor 0 from the same start -- what a plain image-space gradient does with the same 72 angles

    python tools/try_human_target.py ITER LR MODE [MODE ...]     MODE: ref | fresh | noquant | fresh+noquant | clean | prb
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
its = int(sys.argv[1]) if len(sys.argv) > 1 else 120
lr = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
modes = sys.argv[3:] or ["ref", "fresh"]
mname = "match_" + os.environ.get("HUMAN_MATCHER", tasks.matcher)
prb_spp = int(os.environ.get("HUMAN_PRB_SPP", 16))
if os.environ.get("HUMAN_SPP"):                      # samples per pixel of the render the matcher sees (the reference: 64)
    tasks.spp = int(os.environ["HUMAN_SPP"])


def ldr(img, quant):
    if quant:
        return to_ldr(img)
    x = img.clamp(0, 1)
    return torch.where(x <= 0.0031308, 12.92 * x, 1.055 * x.clamp_min(1e-12).pow(1 / 2.4) - 0.055).clamp(0, 1)


for mode in modes:
    quant = "noquant" not in mode and mode != "clean"
    scene = tasks.load_scene(dev)
    gts = tasks.gt_scene(dev)
    integ = load_dict({"type": "manifold", "max_depth": tasks.max_depth})
    matcher = Matcher(tasks.match_res, dev)
    opt, apply_t, backward, output = tasks.optim_settings(scene)
    optimizer = torch.optim.Adam(list(opt.values()), lr=lr)
    rep = tasks.resolution // tasks.match_res
    gt_low = resize(ldr(gts.render_primal(sensor=0, seed=0, spp=8192 if mode == "clean" else 512, max_depth=tasks.max_depth), quant), tasks.match_res)
    hist = [output(opt)]
    gt_eval = gts.render_primal(sensor=0, seed=777, spp=256, max_depth=tasks.max_depth)[..., :3]

    def image_mse():
        apply_t(scene, opt)
        return float(((scene.render_primal(sensor=0, seed=778, spp=256, max_depth=tasks.max_depth)[..., :3] - gt_eval) ** 2).mean())
    mse = [image_mse()]
    for it in range(its):
        apply_t(scene, opt)
        if "fresh" in mode:
            gt_low = resize(ldr(gts.render_primal(sensor=0, seed=10_000 + it, spp=512, max_depth=tasks.max_depth), quant), tasks.match_res)
        params = scene.param_grads()
        if mode == "prb":
            if it == 0:
                prb = load_dict({"type": "prb_reparam", "max_depth": tasks.max_depth})
                ref_img = gts.render_primal(sensor=0, seed=0, spp=512, max_depth=tasks.max_depth)[..., :3]
            img = prb.render(scene, sensor=0, seed=it, spp=prb_spp)
            prb.render_backward(scene, params, 2.0 * (img - ref_img) / img.shape[0], sensor=0, seed=it, spp=prb_spp)
        else:
            img = integ.render(scene, sensor=1, seed=it, spp=tasks.spp)
            low = resize(ldr(img[..., :3], quant), tasks.match_res)
            g = getattr(matcher, mname)(low.reshape(-1, 3), gt_low.reshape(-1, 3)).reshape(tasks.match_res, tasks.match_res, 5).repeat(rep, rep, 1)
            integ.render_backward(scene, params, g, sensor=1, seed=it, spp=tasks.spp)
        backward(opt, params)
        for p in opt.values():
            if p.grad is not None:
                p.grad = torch.nan_to_num(p.grad, nan=0.0, posinf=0.0, neginf=0.0)
        optimizer.step()
        hist.append(output(opt))
        if (it + 1) % 4 == 0:
            mse.append(image_mse())
    tp = tasks.target_pose().reshape(-1)
    err = (opt["pose"].detach().cpu().reshape(-1) - tp).reshape(24, 3)
    print("image mse x1e4 (every 4th iteration)", [round(m * 1e4, 1) for m in mse])
    print("final angle error per joint (|.| over the 3 axes, rad):", [round(float(e), 3) for e in err.norm(dim=1)],
          " at the clamp:", int((opt["pose"].detach().abs() >= 0.0999).sum()), "of 72")
    print(f"## mode {mode}: {its} iterations, lr {lr}, {mname}: best {min(hist):.4f} at {hist.index(min(hist))}, last-20 mean {sum(hist[-20:]) / 20:.4f}")
    print("history", [round(h, 4) for h in hist[::4]], flush=True)
