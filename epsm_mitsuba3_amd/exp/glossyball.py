"""Three coloured lights over a glossy plate, as EPSM/exp/glossyball.py sets them (`myemitterr/g/b`: small rectangles of
radiance (R, 0, 0), (0, G, 0), (0, 0, B) whose translations are optimised, glossyball.py:254-273; a roughconductor plate, a
diffuse floor, an `envmap` as fill light, glossyball.py:44-109): each light's highlight on the plate has to be pulled onto its
target position.  The reference's plate.obj and cyclorama .exr are not part of the repository: the curved plate of exp/plate.py
and the synthetic sky of exp/highlight.py stand in; positions are scaled to that plate.  Three sensors as in exp/plate.py."""
import numpy as np
import torch

from . import plate as _plate
from .highlight import sky
from ..scene import Scene, rotate

it, spp, resolution, thres, max_depth, match_res = _plate.it, _plate.spp, _plate.resolution, _plate.thres, _plate.max_depth, _plate.match_res

_LIGHTS = {"myemitterr": ((-0.7, -0.3, 3.0), (9.0, 0.0, 0.0)), "myemitterg": ((0.1, 0.8, 3.0), (0.0, 9.0, 0.0)),
           "myemitterb": ((0.8, -0.2, 3.0), (0.0, 0.0, 9.0))}
_TARGET = {"myemitterr": np.array([0.5, 0.3, 0.0]), "myemitterg": np.array([-0.4, -0.5, 0.0]), "myemitterb": np.array([-0.3, 0.5, 0.0])}


def load_scene(device="cuda", shifts=None):
    pv, pf = _plate._plate_vertices()
    fv, ff = _plate._quad(0.0, 4.0)
    d = {"type": "scene", "sensor0": _plate._sensor(resolution, spp), "sensor1": _plate._sensor(resolution, spp),
         "sensor2": _plate._sensor(match_res, 8),
         "plate": {"type": "mesh", "vertices": pv, "faces": pf,
                   "bsdf": {"type": "roughconductor", "material": "Al", "distribution": "ggx", "alpha": 0.02}},
         "floor": {"type": "mesh", "vertices": fv, "faces": ff, "face_normals": True,
                   "bsdf": {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.5, 0.5, 0.5]}}},
         "sky": {"type": "envmap", "bitmap": sky(level=0.15), "to_world": rotate([1, 0, 0], 90.0)}}
    for name, (pos, rad) in _LIGHTS.items():
        lv, lf = _plate._quad(0.0, 0.25)
        lv = lv + np.asarray(pos) + (np.asarray(shifts[name]) if shifts else 0.0)
        d[name] = {"type": "mesh", "vertices": lv, "faces": lf[:, ::-1], "face_normals": True,
                   "emitter": {"type": "area", "radiance": {"type": "rgb", "value": list(rad)}}}
    return Scene.from_dict(d, device=device)


def gt_scene(device="cuda"):
    return load_scene(device, _TARGET)


def optim_settings(scene):
    """-> (opt, apply_transformation, backward, output) as in exp/plate.py; one in-plane translation per light."""
    init = {n: scene.vertex_positions(n).clone() for n in _LIGHTS}
    opt = {n: torch.zeros(3, device=scene.device, requires_grad=True) for n in _LIGHTS}
    for n in _LIGHTS:
        scene.attach(n, positions=True)

    def apply_transformation(scene_, opt_):
        for n in _LIGHTS:
            scene_.set_vertex_positions(n, init[n] + opt_[n].detach())

    def backward(opt_, params):
        for n in _LIGHTS:
            g = params.mesh_pos(n).sum(dim=0)
            g[2] = 0                               # glossyball.py:266 keeps the lights in their plane
            opt_[n].grad = g.clone()

    def output(opt_):
        return float(sum((opt_[n].detach().cpu() - torch.tensor(_TARGET[n], dtype=torch.float32))[:2].norm() for n in _LIGHTS) / len(_LIGHTS))

    return opt, apply_transformation, backward, output
