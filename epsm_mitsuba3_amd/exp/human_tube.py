"""Config 5 (EPSM/optim_human.py + exp/human.py): the vertices of a skinned mesh are produced by a torch
module from pose parameters; EPSM hands back PER-VERTEX position gradients, which are chained into the module
with ``loss = sum(verts * grad); loss.backward()`` (optim_human.py:118-121) and Adam updates the pose
(``lr = 0.01``, pose clamped every iteration, optim_human.py:57,95-96).  The SMPL assets of the reference are
not available, so the "human" here is a standing three-bone tube with linear-blend skinning (same interface:
``model.gen_mesh(pose, shape) -> (1, V, 3)``); like the reference scene it is diffuse, lit by a tiny far light,
seen directly and through its shadow on the floor, with ``max_depth = 3`` -- the gradients arrive through
``si_follow.p * diffuse_grad[0]`` (the figure itself, epsm.py:561-562) and through the occluder term
(its shadow, epsm.py:609-620)."""
import math

import numpy as np
import torch

from ..scene import Scene, look_at

it = 60
spp = 16
resolution = 64
thres = 10000
max_depth = 3
match_res = 32

# axis-angle of the two joints: bends about the viewing axis (sideways in the image).  A bend towards the camera is
# a case the method itself is ambiguous about: the first-hit term can only slide a visible point inside its triangle.
_TARGET_POSE = torch.tensor([[0.0, 0.40, 0.0], [0.0, -0.55, 0.0]])
POSE_CLAMP = 0.6                                                          # optim_human.py:96 clamps SMPL's pose to +-0.1


def _rodrigues(r: torch.Tensor) -> torch.Tensor:
    """(3,) axis-angle -> (3,3), differentiable at 0."""
    th = torch.sqrt((r * r).sum() + 1e-12)
    k = r / th
    K = torch.stack([torch.stack([torch.zeros_like(th), -k[2], k[1]]),
                     torch.stack([k[2], torch.zeros_like(th), -k[0]]),
                     torch.stack([-k[1], k[0], torch.zeros_like(th)])])
    return torch.eye(3, dtype=r.dtype, device=r.device) + torch.sin(th) * K + (1 - torch.cos(th)) * (K @ K)


class SkinnedTube:
    """Stand-in for exp/human.py's ``SMPL`` wrapper: ``gen_mesh(pose_params, shape_params)``."""

    def __init__(self, device="cpu", rings=40, sectors=20, radius=0.16, z0=0.15, z1=2.1):
        self.device = torch.device(device)
        z = np.linspace(z0, z1, rings)
        a = np.linspace(0, 2 * np.pi, sectors, endpoint=False)
        v = np.stack([radius * np.cos(a)[None, :].repeat(rings, 0), radius * np.sin(a)[None, :].repeat(rings, 0),
                      z[:, None].repeat(sectors, 1)], -1).reshape(-1, 3)
        caps = np.array([[0, 0, z0], [0, 0, z1]])
        self.rest = torch.tensor(np.concatenate([v, caps]), dtype=torch.float32, device=self.device)
        f = []
        for j in range(rings - 1):
            for i in range(sectors):
                p, q = j * sectors + i, j * sectors + (i + 1) % sectors
                f += [[p, q, q + sectors], [p, q + sectors, p + sectors]]
        nb, nt = rings * sectors, rings * sectors + 1
        for i in range(sectors):
            f.append([nb, (i + 1) % sectors, i])
            f.append([nt, (rings - 1) * sectors + i, (rings - 1) * sectors + (i + 1) % sectors])
        self.faces = np.array(f)
        self.joints = torch.tensor([[0, 0, 0.8], [0, 0, 1.45]], dtype=torch.float32, device=self.device)
        # smooth skinning weights of the three bones along z
        zz = self.rest[:, 2]
        s1 = torch.sigmoid((zz - 0.8) / 0.12)
        s2 = torch.sigmoid((zz - 1.45) / 0.12)
        self.weights = torch.stack([1 - s1, s1 * (1 - s2), s1 * s2], -1)

    def gen_mesh(self, pose_params: torch.Tensor, shape_params=None) -> torch.Tensor:
        """pose_params (1, 6) or (2, 3) axis-angles of the two joints -> (1, V, 3)."""
        pose = pose_params.reshape(2, 3).to(self.device)
        R1, R2 = _rodrigues(pose[0]), _rodrigues(pose[1])
        j1, j2 = self.joints[0], self.joints[1]
        v = self.rest
        b0 = v
        b1 = (v - j1) @ R1.T + j1
        j2w = (j2 - j1) @ R1.T + j1                                   # joint 2 carried by bone 1
        b2 = ((v - j2) @ R2.T) @ R1.T + j2w
        w = self.weights
        return (b0 * w[:, :1] + b1 * w[:, 1:2] + b2 * w[:, 2:3])[None]


def _quad(z, half):
    v = np.array([[-half, -half, z], [half, -half, z], [half, half, z], [-half, half, z]], float)
    return v, np.array([[0, 1, 2], [0, 2, 3]])


def _sensor(res, spp_):
    return {"type": "perspective", "fov": 55, "near_clip": 0.01, "far_clip": 100.0,
            "to_world": look_at([0.6, -4.6, 3.2], [0.3, 0.4, 0.7], [0, 0, 1]),
            "film": {"type": "hdrfilm", "width": res, "height": res, "rfilter": {"type": "gaussian"}},
            "sampler": {"type": "independent", "sample_count": spp_}}


def load_scene(device="cuda", pose=None):
    model_ = SkinnedTube("cpu")
    verts = model_.gen_mesh(torch.zeros(2, 3) if pose is None else pose)[0].numpy().astype(np.float64)
    fv, ff = _quad(0.0, 6.0)
    lv, lf = _quad(9.0, 0.05)
    lv = lv + np.array([-2.5, -3.0, 0.0])
    d = {"type": "scene", "sensor0": _sensor(resolution, spp), "sensor1": _sensor(resolution, spp),
         "sensor2": _sensor(match_res, 8),
         "floor": {"type": "mesh", "vertices": fv, "faces": ff, "face_normals": True,
                   "bsdf": {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.5, 0.5, 0.5]}}},
         "human": {"type": "mesh", "vertices": verts, "faces": model_.faces,
                   "bsdf": {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.8, 0.5, 0.5]}}},
         "light": {"type": "mesh", "vertices": lv, "faces": lf[:, ::-1], "face_normals": True,
                   "emitter": {"type": "area", "radiance": {"type": "rgb", "value": 60000.0}}}}
    return Scene.from_dict(d, device=device)


def gt_scene(device="cuda"):
    return load_scene(device, _TARGET_POSE)


def optim_settings(scene):
    """The steps of optim_human.py:92-122 in the shape ``optim.run`` drives: ``apply_transformation`` =
    clamp pose, gen_mesh, params['human.vertex_positions'] = verts, params.update(); ``backward`` =
    x = dr.grad(optim_vert); NaN -> 0; loss = sum(verts * x); loss.backward()."""
    from ..optim import chain_vertex_grads
    model = SkinnedTube(scene.device)
    opt = {"pose": torch.zeros((1, 6), device=scene.device, requires_grad=True)}
    scene.attach("human", positions=True)
    state = {}

    def apply_transformation(scene_, opt_):
        with torch.no_grad():
            opt_["pose"].clamp_(-POSE_CLAMP, POSE_CLAMP)
            # free: the sideways bend of each joint.  A twist about the tube's own axis is invisible and a bend
            # towards the camera is ambiguous for the method (see _TARGET_POSE): both only collect drift.
            opt_["pose"][:, 2::3] = 0
            opt_["pose"][:, 0::3] = 0
        state["verts"] = model.gen_mesh(opt_["pose"])[0]
        scene_.set_vertex_positions("human", state["verts"].detach())

    def backward(opt_, params):
        x = torch.nan_to_num(params.mesh_pos("human"), nan=0.0)
        opt_["pose"].grad = None
        chain_vertex_grads(state["verts"], x)

    def output(opt_):
        return float((opt_["pose"].detach().cpu().reshape(2, 3) - _TARGET_POSE).norm())

    return opt, apply_transformation, backward, output
