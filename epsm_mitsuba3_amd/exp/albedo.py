"""Colour experiment for the second phase of the hybrid scheme (``prb`` / ``prb_reparam`` / ``*_hybrid``,
EPSM/optim.py:87-94, 113-119, 137-141): recover the reflectance of a diffuse floor and of a wall that lights it
indirectly from a target image, with the colour adjoint of integrators.PRBIntegrator and the L2 image loss of the
reference's non-EPSM branch.  (The reference's experiments optimise geometry in this phase through prb_reparam's warp
field, which is not built here; colour parameters are what the phase can move.)"""
import numpy as np
import torch

from ..scene import Scene, look_at

it = 60
spp = 32
resolution = 32
thres = 3                 # *_hybrid: the manifold phase (which has nothing to move here) hands over after 3 iterations
max_depth = 4
match_res = 16

_TARGET = {"floor": [0.7, 0.3, 0.2], "wall": [0.2, 0.6, 0.7]}
_START = {"floor": [0.4, 0.4, 0.4], "wall": [0.4, 0.4, 0.4]}


def _sensor(res, n):
    return {"type": "perspective", "fov": 40, "near_clip": 0.01, "far_clip": 100.0,
            "to_world": look_at([0.0, -3.5, 1.6], [0, 0.5, 0.5], [0, 0, 1]),
            "film": {"type": "hdrfilm", "width": res, "height": res, "rfilter": {"type": "gaussian"}},
            "sampler": {"type": "independent", "sample_count": n}}


def load_scene(device="cuda", colors=None):
    c = colors or _START
    fv = np.array([[-2, -2, 0], [2, -2, 0], [2, 2, 0], [-2, 2, 0]], float)
    ff = np.array([[0, 1, 2], [0, 2, 3]])
    wv = np.array([[-2, 2, 0], [2, 2, 0], [2, 2, 2.5], [-2, 2, 2.5]], float)
    wf = np.array([[0, 2, 1], [0, 3, 2]])
    lv = np.array([[-0.4, -0.4, 2.2], [0.4, -0.4, 2.2], [0.4, 0.4, 2.2], [-0.4, 0.4, 2.2]], float)
    d = {"type": "scene", "sensor0": _sensor(resolution, spp), "sensor1": _sensor(resolution, spp), "sensor2": _sensor(match_res, 8),
         "floor": {"type": "mesh", "vertices": fv, "faces": ff, "face_normals": True,
                   "bsdf": {"type": "diffuse", "reflectance": {"type": "rgb", "value": list(c["floor"])}}},
         "wall": {"type": "mesh", "vertices": wv, "faces": wf, "face_normals": True,
                  "bsdf": {"type": "twosided", "bsdf": {"type": "diffuse", "reflectance": {"type": "rgb", "value": list(c["wall"])}}}},
         "light": {"type": "mesh", "vertices": lv, "faces": ff[:, ::-1], "face_normals": True,
                   "emitter": {"type": "area", "radiance": {"type": "rgb", "value": 25.0}}}}
    sc = Scene.from_dict(d, device=device)
    sc.tracer = "mega"
    return sc


def gt_scene(device="cuda"):
    return load_scene(device, _TARGET)


def optim_settings(scene):
    slots = {"floor": scene.attach_color("floor.bsdf"), "wall": scene.attach_color("wall.bsdf")}
    opt = {k: torch.tensor(_START[k], device=scene.device, requires_grad=True) for k in slots}

    def apply_transformation(scene_, opt_):
        for k, slot in slots.items():
            scene_.set_color(slot, opt_[k].detach().clamp(0.02, 0.98).tolist())

    def backward(opt_, params):
        for k, slot in slots.items():
            opt_[k].grad = params.color[slot].clone() if params.C else torch.zeros_like(opt_[k])

    def output(opt_):
        return float(sum((opt_[k].detach().cpu() - torch.tensor(_TARGET[k])).norm() for k in slots))

    return opt, apply_transformation, backward, output
