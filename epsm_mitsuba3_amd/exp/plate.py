"""A specular plate under an area light: the highlight seen by the camera has to be moved onto its
target position by translating the LIGHT (the shape of EPSM/exp/highlight.py, whose `li` parameters
translate the emitter meshes: roughconductor GGX alpha 0.01-0.02, 'Al').  Like the reference, the
gradient reaches the emitter through `light_grad` (epsm.py:622-627).  Note: the reference's gradient
has no term for a specular surface moving tangentially under the path (dldp is zeroed unless the
first hit is diffuse, epsm.py:791), so translating the plate itself is not a case it can optimise.  Three sensors like the reference's scenes: 0 = PRB-style, 1 = primal EPSM,
2 = low-resolution backward sensor (exp/shadow.py:27-45,117-154)."""
import numpy as np
import torch

from ..scene import Scene, look_at, rotate

it = 40
spp = 16
resolution = 64
thres = 10000
max_depth = 3
match_res = 32

_TARGET_SHIFT = np.array([0.6, 0.4, 0.0])


def _quad(z, half):
    v = np.array([[-half, -half, z], [half, -half, z], [half, half, z], [-half, half, z]], float)
    return v, np.array([[0, 1, 2], [0, 2, 3]])


def _sensor(res, spp_):
    return {"type": "perspective", "fov": 45, "near_clip": 0.01, "far_clip": 100.0,
            "to_world": look_at([0.0, -3.0, 2.5], [0.0, 0.0, 0.3], [0, 0, 1]),
            "film": {"type": "hdrfilm", "width": res, "height": res, "rfilter": {"type": "gaussian"}},
            "sampler": {"type": "independent", "sample_count": spp_}}


def _plate_vertices():
    # a gently curved 9x9 grid so that the highlight is a compact blob
    n = 9
    u = np.linspace(-0.9, 0.9, n)
    X, Y = np.meshgrid(u, u, indexing="xy")
    Z = 0.3 - 0.12 * (X ** 2 + Y ** 2)
    v = np.stack([X, Y, Z], -1).reshape(-1, 3)
    v = (rotate([1, 0, 0], 8.0)[:3, :3] @ v.T).T
    f = []
    for j in range(n - 1):
        for i in range(n - 1):
            a = j * n + i
            f += [[a, a + 1, a + n + 1], [a, a + n + 1, a + n]]
    return v, np.array(f)


def load_scene(device="cuda", shift=(0.0, 0.0, 0.0)):
    pv, pf = _plate_vertices()
    fv, ff = _quad(0.0, 4.0)
    lv, lf = _quad(3.0, 0.35)
    lv = lv + np.asarray(shift)
    d = {"type": "scene", "sensor0": _sensor(resolution, spp), "sensor1": _sensor(resolution, spp),
         "sensor2": _sensor(match_res, 8),
         "plate": {"type": "mesh", "vertices": pv, "faces": pf,
                   "bsdf": {"type": "roughconductor", "material": "Al", "distribution": "ggx", "alpha": 0.02}},
         "floor": {"type": "mesh", "vertices": fv, "faces": ff, "face_normals": True,
                   "bsdf": {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.2, 0.2, 0.2]}}},
         "light": {"type": "mesh", "vertices": lv, "faces": lf[:, ::-1], "face_normals": True,
                   "emitter": {"type": "area", "radiance": {"type": "rgb", "value": 12.0}}}}
    return Scene.from_dict(d, device=device)


def gt_scene(device="cuda"):
    return load_scene(device, _TARGET_SHIFT)


def optim_settings(scene):
    """-> (opt, apply_transformation, backward, output), the torch counterpart of the reference's
    ``optim_settings()`` (e.g. exp/bathroom.py:12-42): ``opt`` holds the optimised leaves,
    ``apply_transformation`` writes them into the scene, ``backward`` chains the vertex gradients of
    ``ParamGrads`` into ``opt[...].grad``, ``output`` reports the parameter error."""
    init = scene.vertex_positions("light").clone()
    opt = {"trans": torch.zeros(3, device=scene.device, requires_grad=True)}
    scene.attach("light", positions=True)

    def apply_transformation(scene_, opt_):
        scene_.set_vertex_positions("light", init + opt_["trans"].detach())

    def backward(opt_, params):
        g = params.mesh_pos("light").sum(dim=0)
        g[2] = 0                                   # the experiment optimises the in-plane translation
        opt_["trans"].grad = g.clone()

    def output(opt_):
        return float((opt_["trans"].detach().cpu() - torch.tensor(_TARGET_SHIFT, dtype=torch.float32))[:2].norm())

    return opt, apply_transformation, backward, output
