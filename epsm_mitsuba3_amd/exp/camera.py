"""The camera's position as the unknown (the shape of EPSM/exp/bedroom.py:18-34, whose `trans2` translates
`PerspectiveCamera.to_world`): the gradient arrives through the ray origins alone -- epsm.py:260-261, `dr.backward(ray.o * -grad_d)`,
the sum over the paths that `render_backward` leaves in `ParamGrads.cam_origin`.  The reference's bedroom scene (an .xml with its
assets) is not part of the repository: the plate scene of exp/plate.py with two coloured blocks on the floor stands in; all three
sensors move together."""
import numpy as np
import torch

from . import plate as _plate
from ..scene import Scene

it, spp, resolution, thres, max_depth, match_res = 60, _plate.spp, _plate.resolution, _plate.thres, _plate.max_depth, _plate.match_res

_TARGET_SHIFT = np.array([0.35, 0.0, 0.25])


def _block(center, half, colour):
    c, h = np.asarray(center, float), np.asarray(half, float)
    v = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], float) * h + c
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    f = []
    for a, b, c_, d in quads:
        f += [[a, b, c_], [a, c_, d]]
    return {"type": "mesh", "vertices": v, "faces": np.array(f), "face_normals": True,
            "bsdf": {"type": "diffuse", "reflectance": {"type": "rgb", "value": list(colour)}}}


def load_scene(device="cuda", shift=(0.0, 0.0, 0.0)):
    pv, pf = _plate._plate_vertices()
    fv, ff = _plate._quad(0.0, 4.0)
    lv, lf = _plate._quad(3.0, 0.6)
    d = {"type": "scene", "sensor0": _plate._sensor(resolution, spp), "sensor1": _plate._sensor(resolution, spp),
         "sensor2": _plate._sensor(match_res, 8),
         "plate": {"type": "mesh", "vertices": pv, "faces": pf,
                   "bsdf": {"type": "roughconductor", "material": "Al", "distribution": "ggx", "alpha": 0.05}},
         "floor": {"type": "mesh", "vertices": fv, "faces": ff, "face_normals": True,
                   "bsdf": {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.3, 0.3, 0.3]}}},
         "red": _block((-1.0, 0.6, 0.25), (0.25, 0.25, 0.25), (0.8, 0.15, 0.1)),
         "blue": _block((1.1, 0.2, 0.2), (0.2, 0.3, 0.2), (0.1, 0.2, 0.8)),
         "light": {"type": "mesh", "vertices": lv, "faces": lf[:, ::-1], "face_normals": True,
                   "emitter": {"type": "area", "radiance": {"type": "rgb", "value": 10.0}}}}
    sc = Scene.from_dict(d, device=device)
    move_cameras(sc, [s.to_world.copy() for s in sc.sensors], np.asarray(shift, float))
    return sc


def move_cameras(scene, init, t):
    for s, m in zip(scene.sensors, init):
        w = m.copy(); w[:3, 3] = m[:3, 3] + t
        s.to_world = w


def gt_scene(device="cuda"):
    return load_scene(device, _TARGET_SHIFT)


def optim_settings(scene):
    init = [s.to_world.copy() for s in scene.sensors]
    opt = {"trans": torch.zeros(3, device=scene.device, requires_grad=True)}

    def apply_transformation(scene_, opt_):
        move_cameras(scene_, init, opt_["trans"].detach().cpu().double().numpy())

    def backward(opt_, params):
        g = params.cam_origin.clone()
        g[1] = 0                                   # (the depth direction of this view is barely determined by the image)
        opt_["trans"].grad = g

    def output(opt_):
        d = opt_["trans"].detach().cpu() - torch.tensor(_TARGET_SHIFT, dtype=torch.float32)
        return float(d[[0, 2]].norm())

    return opt, apply_transformation, backward, output
