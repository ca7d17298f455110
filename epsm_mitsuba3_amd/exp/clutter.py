"""A room-sized scene with many triangles for timing the whole pipeline on traced records: a diffuse floor, `n`
tessellated spheres (1 280 triangles each; every third one a rough conductor, GGX alpha 0.05, 'Al', the others
diffuse) and an area light above -- the stand-in for the reference's bathroom asset, which is not in its
repository (EPSM/exp/bathroom.py loads `scenes/bathroom/*.obj`).  Three sensors like the reference's scenes
(0 = PRB-style, 1 = primal EPSM, 2 = backward sensor; exp/shadow.py:27-45,117-154), all at `res`."""
import numpy as np

from ..scene import Scene, look_at

max_depth = 4


def icosphere(sub: int):
    t = (1 + 5 ** 0.5) / 2
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], float)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6],
                  [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7],
                  [9, 8, 1]])
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    for _ in range(sub):
        cache, nf, vl = {}, [], list(v)

        def mid(a, b):
            k = (min(a, b), max(a, b))
            if k not in cache:
                m = (vl[a] + vl[b]) / 2
                vl.append(m / np.linalg.norm(m))
                cache[k] = len(vl) - 1
            return cache[k]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [[a, ab, ca], [b, bc, ab], [c, ca, bc], [ab, bc, ca]]
        v, f = np.array(vl), np.array(nf)
    return v, f


def scene_dict(n_spheres: int = 100, res: int = 512, spp: int = 16, seed: int = 0) -> dict:
    sv, sf = icosphere(3)                                             # 642 vertices, 1 280 triangles
    rng = np.random.default_rng(seed)
    cam = {"type": "perspective", "fov": 50, "to_world": look_at([0, -6, 4], [0, 0, 0.5], [0, 0, 1]),
           "film": {"type": "hdrfilm", "width": res, "height": res, "rfilter": {"type": "gaussian"}},
           "sampler": {"type": "independent", "sample_count": spp}}
    d = {"type": "scene", "sensor0": cam, "sensor1": dict(cam), "sensor2": dict(cam)}
    fv = np.array([[-6, -6, 0], [6, -6, 0], [6, 6, 0], [-6, 6, 0]], float)
    ff = np.array([[0, 1, 2], [0, 2, 3]])
    d["floor"] = {"type": "mesh", "vertices": fv, "faces": ff, "face_normals": True, "bsdf": {"type": "diffuse"}}
    for i in range(n_spheres):
        c = np.array([rng.uniform(-4, 4), rng.uniform(-4, 4), rng.uniform(0.3, 2.0)])
        r = rng.uniform(0.15, 0.35)
        d[f"s{i}"] = {"type": "mesh", "vertices": sv * r + c, "faces": sf,
                      "bsdf": {"type": "roughconductor", "material": "Al", "distribution": "ggx", "alpha": 0.05}
                      if i % 3 == 0 else {"type": "diffuse"}}
    lv = np.array([[-1, -1, 6], [1, -1, 6], [1, 1, 6], [-1, 1, 6]], float)
    d["light"] = {"type": "mesh", "vertices": lv, "faces": ff[:, ::-1], "face_normals": True,
                  "emitter": {"type": "area", "radiance": {"type": "rgb", "value": 20.0}}}
    return d


def load_scene(device="cuda", n_spheres: int = 100, res: int = 512, spp: int = 16, seed: int = 0) -> Scene:
    return Scene.from_dict(scene_dict(n_spheres, res, spp, seed), device=device)
