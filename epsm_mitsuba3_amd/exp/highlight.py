"""The plate experiment (exp/plate.py) lit as EPSM/exp/highlight.py lights its scenes: the small area light whose highlight is
matched PLUS an environment map as fill light (highlight.py:218-222, 412-416: `envmap`, drachenfels_cellar_1k.exr -- not part of
the repository; a smooth synthetic sky stands in).  Paths that sample the environment log a far point as their emitter sample
and add no emitter rows (epsm.py:622-627 finds no surface to follow); the highlight of the area light still has to pull the light
onto its target."""
import math

import numpy as np

from . import plate as _plate
from ..scene import Scene, rotate

it, spp, resolution, thres, max_depth, match_res = _plate.it, _plate.spp, _plate.resolution, _plate.thres, _plate.max_depth, _plate.match_res
optim_settings = _plate.optim_settings


def sky(H=16, W=32, level=0.25):
    """A dim sky, brighter towards +z (the plate scene's up) and warmer on one side."""
    j, i = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    th, ph = math.pi * j / (H - 1), 2 * math.pi * (i + 0.5) / W
    up = 0.5 + 0.5 * np.cos(th)
    a = np.stack([up * (1.0 + 0.3 * np.cos(ph)), up, up * (1.0 - 0.3 * np.cos(ph))], -1)
    return (level * a).astype(np.float32)


def load_scene(device="cuda", shift=(0.0, 0.0, 0.0)):
    pv, pf = _plate._plate_vertices()
    fv, ff = _plate._quad(0.0, 4.0)
    lv, lf = _plate._quad(3.0, 0.35)
    lv = lv + np.asarray(shift)
    d = {"type": "scene", "sensor0": _plate._sensor(resolution, spp), "sensor1": _plate._sensor(resolution, spp),
         "sensor2": _plate._sensor(match_res, 8),
         "plate": {"type": "mesh", "vertices": pv, "faces": pf,
                   "bsdf": {"type": "roughconductor", "material": "Al", "distribution": "ggx", "alpha": 0.02}},
         "floor": {"type": "mesh", "vertices": fv, "faces": ff, "face_normals": True,
                   "bsdf": {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.2, 0.2, 0.2]}}},
         "light": {"type": "mesh", "vertices": lv, "faces": lf[:, ::-1], "face_normals": True,
                   "emitter": {"type": "area", "radiance": {"type": "rgb", "value": 12.0}}},
         # the map's pole (+y of its own frame) turned onto the scene's up axis +z
         "sky": {"type": "envmap", "bitmap": sky(), "to_world": rotate([1, 0, 0], 90.0)}}
    return Scene.from_dict(d, device=device)


def gt_scene(device="cuda"):
    return load_scene(device, _plate._TARGET_SHIFT)
