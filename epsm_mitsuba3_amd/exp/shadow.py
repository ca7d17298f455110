"""An occluder between a small area light and a diffuse floor: its shadow has to be moved onto the target
position by translating the occluder (the shape of EPSM/exp/shadow.py: 400 little spheres under a 1 cm light,
max_depth = 2, each translated in the plane).  The gradient reaches the occluder through the reference's
occluder term (epsm.py:609-620, integrators with max_depth <= 3): the first hit is the diffuse floor, so
diffuse_grad[0] = dldp (epsm.py:791-792) is the motion the matcher asks of the floor point, and the closest hit
of the ray towards the emitter sample -- the occluder -- receives it scaled by dis = |light - occluder| /
|light - floor point|.  Three sensors like the reference's scenes (exp/shadow.py:27-45,117-154)."""
import numpy as np
import torch

from ..scene import Scene, look_at

it = 40
spp = 16
resolution = 64
thres = 10000
max_depth = 2
match_res = 32

_TARGET_SHIFT = np.array([0.5, -0.35, 0.0])


def _quad(z, half):
    v = np.array([[-half, -half, z], [half, -half, z], [half, half, z], [-half, half, z]], float)
    return v, np.array([[0, 1, 2], [0, 2, 3]])


def _disc(z, radius, n=24):
    a = np.linspace(0, 2 * np.pi, n, endpoint=False)
    v = np.concatenate([[[0.0, 0.0, z]], np.stack([radius * np.cos(a), radius * np.sin(a), np.full(n, z)], -1)])
    f = np.array([[0, 1 + i, 1 + (i + 1) % n] for i in range(n)])
    return v, f


def _sensor(res, spp_, sample_border=False):
    return {"type": "perspective", "fov": 50, "near_clip": 0.01, "far_clip": 100.0,
            "to_world": look_at([0.0, -3.2, 3.0], [0.0, 0.0, 0.0], [0, 0, 1]),
            "film": {"type": "hdrfilm", "width": res, "height": res, "rfilter": {"type": "gaussian"}, "sample_border": sample_border},
            "sampler": {"type": "independent", "sample_count": spp_}}


def load_scene(device="cuda", shift=(0.0, 0.0, 0.0)):
    fv, ff = _quad(0.0, 4.0)
    ov, of = _disc(1.0, 0.45)
    ov = ov + np.asarray(shift)
    lv, lf = _quad(4.0, 0.04)
    # sensor 0 (primal / prb_reparam) samples the film's border, the manifold integrators' sensors do not (exp/shadow.py:38,128,147)
    d = {"type": "scene", "sensor0": _sensor(resolution, spp, True), "sensor1": _sensor(resolution, spp),
         "sensor2": _sensor(match_res, 8),
         "floor": {"type": "mesh", "vertices": fv, "faces": ff, "face_normals": True,
                   "bsdf": {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.5, 0.5, 0.5]}}},
         "occluder": {"type": "mesh", "vertices": ov, "faces": of, "face_normals": True,
                      "bsdf": {"type": "twosided", "bsdf": {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.5, 0.5, 0.5]}}}},
         "light": {"type": "mesh", "vertices": lv, "faces": lf[:, ::-1], "face_normals": True,
                   "emitter": {"type": "area", "radiance": {"type": "rgb", "value": 3000.0}}}}
    return Scene.from_dict(d, device=device)


def gt_scene(device="cuda"):
    return load_scene(device, _TARGET_SHIFT)


def optim_settings(scene):
    init = scene.vertex_positions("occluder").clone()
    opt = {"trans": torch.zeros(3, device=scene.device, requires_grad=True)}
    scene.attach("occluder", positions=True)

    def apply_transformation(scene_, opt_):
        scene_.set_vertex_positions("occluder", init + opt_["trans"].detach())

    def backward(opt_, params):
        g = params.mesh_pos("occluder").sum(dim=0)
        g[2] = 0                                   # exp/shadow.py:250 pins one axis too (`opt[obj][1] = 0`)
        opt_["trans"].grad = g.clone()

    def output(opt_):
        return float((opt_["trans"].detach().cpu() - torch.tensor(_TARGET_SHIFT, dtype=torch.float32))[:2].norm())

    return opt, apply_transformation, backward, output
