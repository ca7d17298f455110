"""The stand-in for the reference's `manifold_caustic` box experiments (EPSM/exp/cornellbox.py, egg.py: EPSM/all.sh:7,11 -- their
assets are not in the repository) and the shape of BASELINE.json configs[0], "single glass-sphere caustic": an open diffuse box
(floor + back wall), a glass SPHERE with vertex normals above the floor, a small area light above it.  The camera looks at the
floor, which receives the light's caustic through the sphere: camera -> diffuse floor -> two refractions on a curved, smoothly
shaded surface -> area light.  The light is translated; its gradient arrives through `diffuse_grad` of the chain's end point
(epsm.py:1178-1184, 561-562), the chain's Jacobian goes through the interpolated normals of two refracting vertices."""
import numpy as np
import torch

from ..scene import Scene, look_at
from .clutter import icosphere

it = 60
spp = 32
resolution = 64
thres = 10000
max_depth = 5
match_res = 32

_TARGET_SHIFT = np.array([0.35, 0.25, 0.0])


def _quad(z, half, up=True):
    v = np.array([[-half, -half, z], [half, -half, z], [half, half, z], [-half, half, z]], float)
    f = np.array([[0, 1, 2], [0, 2, 3]])
    return v, (f if up else f[:, ::-1])


def _sensor(res, spp_):
    return {"type": "perspective", "fov": 50, "near_clip": 0.01, "far_clip": 100.0,
            "to_world": look_at([0.0, -3.4, 1.6], [0.0, 0.1, 0.0], [0, 0, 1]),
            "film": {"type": "hdrfilm", "width": res, "height": res, "rfilter": {"type": "gaussian"}},
            "sampler": {"type": "independent", "sample_count": spp_}}


def load_scene(device="cuda", shift=(0.0, 0.0, 0.0)):
    fv, ff = _quad(0.0, 4.0)
    wv = np.array([[-4, 3.0, 0], [4, 3.0, 0], [4, 3.0, 4], [-4, 3.0, 4]], float)
    wf = np.array([[0, 2, 1], [0, 3, 2]])
    sv, sf = icosphere(3)                                              # 642 vertices, 1 280 triangles, unit sphere
    lv, lf = _quad(3.2, 0.25, up=False)
    glass = {"type": "dielectric", "int_ior": 1.5, "ext_ior": 1.0}
    d = {"type": "scene", "sensor0": _sensor(resolution, spp), "sensor1": _sensor(resolution, spp),
         "sensor2": _sensor(match_res, 8),
         "floor": {"type": "mesh", "vertices": fv, "faces": ff, "face_normals": True,
                   "bsdf": {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.7, 0.7, 0.7]}}},
         "wall": {"type": "mesh", "vertices": wv, "faces": wf, "face_normals": True,
                  "bsdf": {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.5, 0.5, 0.6]}}},
         "ball": {"type": "mesh", "vertices": sv * 0.7 + np.array([0.0, 0.0, 1.3]), "normals": sv, "faces": sf, "bsdf": glass},
         "light": {"type": "mesh", "vertices": lv + np.asarray(shift), "faces": lf, "face_normals": True,
                   "emitter": {"type": "area", "radiance": {"type": "rgb", "value": 60.0}}}}
    return Scene.from_dict(d, device=device)


def gt_scene(device="cuda"):
    return load_scene(device, _TARGET_SHIFT)


def optim_settings(scene):
    init = scene.vertex_positions("light").clone()
    opt = {"trans": torch.zeros(3, device=scene.device, requires_grad=True)}
    scene.attach("light", positions=True)

    def apply_transformation(scene_, opt_):
        scene_.set_vertex_positions("light", init + opt_["trans"].detach())

    def backward(opt_, params):
        g = params.mesh_pos("light").sum(dim=0)
        g[2] = 0
        opt_["trans"].grad = g.clone()

    def output(opt_):
        return float((opt_["trans"].detach().cpu() - torch.tensor(_TARGET_SHIFT, dtype=torch.float32))[:2].norm())

    return opt, apply_transformation, backward, output
