"""Objects seen in a mirror, the configuration EPSM/exp/bathroom.py optimises (``data/bathroom2``: eight decorations
translated in x,y until their mirror images match; `manifold`, max_depth 8): camera -> mirror (delta conductor) ->
diffuse object -> light.  The assets of the reference are not available, so the room is procedural: a floor, a
back wall carrying a large smooth-aluminium mirror, three coloured tiles standing on the floor with their coloured
side towards the mirror (the camera sees their grey backs directly and their colours only in the mirror), an area
light above.  The gradient of an object's translation arrives through ``diffuse_grad[1]`` -- the diffuse end point
x_2 of the specular chain camera -> x_1 (mirror) -> x_2 (epsm.py:906,923-924 -> 561-562) -- and, for its
directly visible silhouette, through ``diffuse_grad[0]``.  Three sensors like the reference's scenes."""
import numpy as np
import torch

from ..scene import Scene, look_at

it = 60
spp = 16
resolution = 64
thres = 10000
max_depth = 4
match_res = 32

OBJECTS = ("tile_red", "tile_green", "tile_blue")
_COLOURS = ([0.85, 0.15, 0.15], [0.15, 0.75, 0.2], [0.2, 0.3, 0.9])
_REST_X = (-1.1, 0.0, 1.1)
# target translations (x, z): sideways and up/down, in the plane of the tiles
_TARGET = np.array([[0.35, 0.25], [-0.3, 0.3], [0.3, -0.2]])


def _rect(x0, x1, z0, z1, y, facing):
    """Vertical rectangle in the plane y = const whose front side faces the direction (0, facing, 0)."""
    v = np.array([[x0, y, z0], [x1, y, z0], [x1, y, z1], [x0, y, z1]], float)
    f = np.array([[0, 1, 2], [0, 2, 3]])              # counter-clockwise seen from -y
    return v, (f if facing < 0 else f[:, ::-1])


def _quad(z, half):
    v = np.array([[-half, -half, z], [half, -half, z], [half, half, z], [-half, half, z]], float)
    return v, np.array([[0, 1, 2], [0, 2, 3]])


def _sensor(res, spp_):
    return {"type": "perspective", "fov": 55, "near_clip": 0.01, "far_clip": 100.0,
            "to_world": look_at([0.0, -4.2, 2.4], [0.0, 2.0, 1.1], [0, 0, 1]),
            "film": {"type": "hdrfilm", "width": res, "height": res, "rfilter": {"type": "gaussian"}},
            "sampler": {"type": "independent", "sample_count": spp_}}


def load_scene(device="cuda", shifts=None):
    shifts = np.zeros((3, 2)) if shifts is None else np.asarray(shifts, dtype=float)
    fv, ff = _quad(0.0, 6.0)
    wv, wf = _rect(-6.0, 6.0, 0.0, 5.0, 3.0, facing=-1)            # back wall
    mv, mf = _rect(-2.6, 2.6, 0.3, 3.2, 2.98, facing=-1)           # mirror, 2 cm in front of it
    lv, lf = _quad(4.8, 0.8)
    lv = lv + np.array([0.0, 0.5, 0.0])
    grey = {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.5, 0.5, 0.5]}}
    d = {"type": "scene", "sensor0": _sensor(resolution, spp), "sensor1": _sensor(resolution, spp),
         "sensor2": _sensor(match_res, 8),
         "floor": {"type": "mesh", "vertices": fv, "faces": ff, "face_normals": True, "bsdf": grey},
         "wall": {"type": "mesh", "vertices": wv, "faces": wf, "face_normals": True, "bsdf": grey},
         "mirror": {"type": "mesh", "vertices": mv, "faces": mf, "face_normals": True,
                    "bsdf": {"type": "conductor", "material": "Al"}},
         "light": {"type": "mesh", "vertices": lv, "faces": lf[:, ::-1], "face_normals": True,
                   "emitter": {"type": "area", "radiance": {"type": "rgb", "value": 18.0}}}}
    for name, col, x, s in zip(OBJECTS, _COLOURS, _REST_X, shifts):
        # coloured side towards the mirror (+y), grey back towards the camera: two coincident sheets 1 mm apart
        cv, cf = _rect(x - 0.4 + s[0], x + 0.4 + s[0], 0.9 + s[1], 1.7 + s[1], 0.6, facing=+1)
        bv, bf = _rect(x - 0.4 + s[0], x + 0.4 + s[0], 0.9 + s[1], 1.7 + s[1], 0.599, facing=-1)
        d[name] = {"type": "mesh", "vertices": cv, "faces": cf, "face_normals": True,
                   "bsdf": {"type": "diffuse", "reflectance": {"type": "rgb", "value": col}}}
        d[name + "_back"] = {"type": "mesh", "vertices": bv, "faces": bf, "face_normals": True, "bsdf": grey}
    return Scene.from_dict(d, device=device)


def gt_scene(device="cuda"):
    return load_scene(device, _TARGET)


def optim_settings(scene):
    """exp/bathroom.py:14-42: one (x, y) translation per object -- here (x, z), the plane the tiles stand in."""
    names = [n for o in OBJECTS for n in (o, o + "_back")]
    init = {n: scene.vertex_positions(n).clone() for n in names}
    opt = {f"trans_{o}": torch.zeros(2, device=scene.device, requires_grad=True) for o in OBJECTS}
    for n in names:
        scene.attach(n, positions=True)

    def apply_transformation(scene_, opt_):
        for o in OBJECTS:
            t = opt_[f"trans_{o}"].detach()
            off = torch.stack([t[0], torch.zeros_like(t[0]), t[1]])
            for n in (o, o + "_back"):
                scene_.set_vertex_positions(n, init[n] + off)

    def backward(opt_, params):
        for o in OBJECTS:
            g = params.mesh_pos(o).sum(dim=0) + params.mesh_pos(o + "_back").sum(dim=0)
            opt_[f"trans_{o}"].grad = torch.stack([g[0], g[2]])

    def output(opt_):
        cur = torch.stack([opt_[f"trans_{o}"].detach().cpu() for o in OBJECTS])
        return float((cur - torch.tensor(_TARGET, dtype=torch.float32)).norm())

    return opt, apply_transformation, backward, output
