"""Outer optimisation loop in the shape of EPSM/optim.py:36-165:

    python -m epsm_mitsuba3_amd.optim METHOD EXP        # METHOD: manifold | manifold_caustic (| *_hybrid)

render (H,W,5) -> tone-map + resize to ``match_res`` -> Sinkhorn matcher -> 5-channel gradient image tiled
back to the film size (optim.py:130-135) -> ``render_backward`` -> chain rule into the optimised leaves
-> NaN scrub (optim.py:143-154) -> Adam step (torch, as the reference's optim_human.py does).
``*_hybrid`` (optim.py:87-94, 113-119): after ``tasks.thres`` iterations the optimiser state is reset and the loop
switches to the ``prb_reparam`` integrator on sensor 0 with the L2 image loss of optim.py:137-141
(``grad_in = 2 (img - ref) / len(img)``, 3 channels): integrators.PRBReparamIntegrator -- vertex positions / normals
through the warp field (csrc/epsm_trace_reparam.h), colour parameters (Scene.attach_color / attach_radiance) through the
colour adjoint.  ``METHOD = prb_reparam`` runs that integrator from the first iteration.
"""
from __future__ import annotations

import importlib
import sys

import torch
import torch.nn.functional as F

from . import load_dict
from .integrators import render_seeds
from .matcher import Matcher


def to_ldr(img: torch.Tensor) -> torch.Tensor:
    """mi.util.convert_to_bitmap + /255 (optim.py:121,131): linear -> sRGB, clamp, 8-bit quantisation."""
    x = img.clamp(0, 1)
    srgb = torch.where(x <= 0.0031308, 12.92 * x, 1.055 * x.clamp_min(1e-12).pow(1 / 2.4) - 0.055)
    return torch.round(srgb.clamp(0, 1) * 255) / 255


def resize(img: torch.Tensor, res: int) -> torch.Tensor:
    """cv2.resize(img, (res,res)) -- bilinear, no anti-aliasing."""
    return F.interpolate(img.permute(2, 0, 1)[None], size=(res, res), mode="bilinear", align_corners=False)[0].permute(1, 2, 0)


def chain_vertex_grads(vertices: torch.Tensor, vertex_grad: torch.Tensor) -> None:
    """EPSM/optim_human.py:118-121: per-vertex position gradients chained into whatever torch module
    produced the vertices (SMPL there): ``loss = sum(verts * grad); loss.backward()``."""
    (vertices * vertex_grad.detach().to(vertices.device, vertices.dtype)).sum().backward()


def run(method: str, exp: str, device="cuda", iterations=None, lr=None, log=print):
    tasks = importlib.import_module(f"epsm_mitsuba3_amd.exp.{exp}")
    lr = getattr(tasks, "lr", 0.02) if lr is None else lr
    # the reference's Matcher has both (utils/matcher.py:51-63 and :76-180); an experiment may name the sort-based one
    match_name = "match_" + getattr(tasks, "matcher", "Sinkhorn")
    thres, integrator2 = 10000, None                                            # optim.py:93-94
    if method.endswith("hybrid"):                                               # optim.py:87-92
        method = method[:-7]
        integrator2 = load_dict({"type": "prb_reparam", "max_depth": tasks.max_depth})
        thres = getattr(tasks, "thres", 10000)
        log(f"hybrid: phase 2 = {integrator2} after {thres} iterations")
    scene = tasks.load_scene(device)
    integrator = load_dict({"type": method, "max_depth": tasks.max_depth})
    sensor_id = 1 if method.startswith("manifold") else 0                      # optim.py:103-106
    gt = tasks.gt_scene(device).render_primal(sensor=0, seed=0, spp=512, max_depth=tasks.max_depth)
    gt_low = resize(to_ldr(gt), tasks.match_res)                               # optim.py:66
    matcher = Matcher(tasks.match_res, device)
    opt, apply_transformation, backward, output = tasks.optim_settings(scene)
    optimizer = torch.optim.Adam(list(opt.values()), lr=lr)
    params = scene.param_grads()
    history = [output(opt)]
    rep = tasks.resolution // tasks.match_res
    for it in range(iterations or tasks.it):
        apply_transformation(scene, opt)                                        # optim.py:112
        phase2 = it >= thres
        if not phase2:
            integ, sid = integrator, sensor_id
        else:
            if it == thres:                                                     # optim.py:116-118: opt.reset(key)
                optimizer = torch.optim.Adam(list(opt.values()), lr=lr)
            integ, sid = integrator2, 0
        seed, seed_grad = render_seeds(it)                                      # util.py:505-513: de-correlated passes
        img = integ.render(scene, sensor=sid, seed=seed, spp=tasks.spp)         # (H,W,5) or (H,W,3)
        params = scene.param_grads() if params.flat.numel() != scene.param_grads().flat.numel() else params
        params.zero_()
        if img.shape[-1] == 5:                                                  # optim.py:130-136
            render_low = resize(to_ldr(img[..., :3]), tasks.match_res)
            grad_ = getattr(matcher, match_name)(render_low.reshape(-1, 3), gt_low.reshape(-1, 3))
            grad = grad_.reshape(tasks.match_res, tasks.match_res, 5).repeat(rep, rep, 1)         # optim.py:133-135
        else:                                                                   # optim.py:137-141: L2 against the reference
            ref = gt if tuple(gt.shape[:2]) == tuple(img.shape[:2]) else resize(gt, img.shape[0])
            grad = 2.0 * (img - ref[..., :3]) / img.shape[0]
        integ.render_backward(scene, params, grad, sensor=sid, seed=seed_grad, spp=tasks.spp)   # dr.backward(img*grad) / dr.backward(loss)
        backward(opt, params)
        for p in opt.values():                                                  # optim.py:143-154
            if p.grad is not None:
                p.grad = torch.nan_to_num(p.grad, nan=0.0, posinf=0.0, neginf=0.0)
        optimizer.step()
        history.append(output(opt))
        log(f"Iteration {it:02d}: error={history[-1]:.5f}")
    return history, opt


if __name__ == "__main__":
    run(sys.argv[1], sys.argv[2])
