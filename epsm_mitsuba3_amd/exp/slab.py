"""A caustic configuration for ``manifold_caustic``: the camera looks at a diffuse floor that receives
light only through a glass slab (camera -> diffuse receiver -> two refractions -> area light), in the
shape of EPSM/exp/glassslab.py (dielectric int_ior 1.5).  The light is translated; its gradient arrives
through ``diffuse_grad`` of the chain's end point (epsm.py:1178-1184, 561-562)."""
import numpy as np
import torch

from ..scene import Scene, look_at

it = 60
spp = 32
resolution = 64
thres = 10000
max_depth = 5
match_res = 32

_TARGET_SHIFT = np.array([0.5, 0.3, 0.0])


def _quad(z, half, up=True):
    v = np.array([[-half, -half, z], [half, -half, z], [half, half, z], [-half, half, z]], float)
    f = np.array([[0, 1, 2], [0, 2, 3]])
    return v, (f if up else f[:, ::-1])


def _sensor(res, spp_):
    return {"type": "perspective", "fov": 50, "near_clip": 0.01, "far_clip": 100.0,
            "to_world": look_at([0.0, -3.2, 0.6], [0.0, 0.2, 0.0], [0, 0, 1]),
            "film": {"type": "hdrfilm", "width": res, "height": res, "rfilter": {"type": "gaussian"}},
            "sampler": {"type": "independent", "sample_count": spp_}}


def load_scene(device="cuda", shift=(0.0, 0.0, 0.0)):
    fv, ff = _quad(0.0, 5.0)
    tv, tf = _quad(1.0, 2.5, up=True)
    bv, bf = _quad(0.8, 2.5, up=False)
    lv, lf = _quad(2.5, 0.5, up=False)
    glass = {"type": "dielectric", "int_ior": 1.5, "ext_ior": 1.0}
    d = {"type": "scene", "sensor0": _sensor(resolution, spp), "sensor1": _sensor(resolution, spp),
         "sensor2": _sensor(match_res, 8),
         "floor": {"type": "mesh", "vertices": fv, "faces": ff, "face_normals": True,
                   "bsdf": {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.7, 0.7, 0.7]}}},
         "slab_top": {"type": "mesh", "vertices": tv, "faces": tf, "face_normals": True, "bsdf": glass},
         "slab_bottom": {"type": "mesh", "vertices": bv, "faces": bf, "face_normals": True, "bsdf": glass},
         "light": {"type": "mesh", "vertices": lv + np.asarray(shift), "faces": lf, "face_normals": True,
                   "emitter": {"type": "area", "radiance": {"type": "rgb", "value": 30.0}}}}
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
