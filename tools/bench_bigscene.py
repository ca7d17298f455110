import sys, time; sys.path.insert(0, ".")
import numpy as np, torch
from epsm_mitsuba3_amd.scene import Scene, look_at
def icosphere(sub):
    t = (1 + 5 ** 0.5) / 2
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t], [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], float)
    f = np.array([[0,11,5],[0,5,1],[0,1,7],[0,7,10],[0,10,11],[1,5,9],[5,11,4],[11,10,2],[10,7,6],[7,1,8],[3,9,4],[3,4,2],[3,2,6],[3,6,8],[3,8,9],[4,9,5],[2,4,11],[6,2,10],[8,6,7],[9,8,1]])
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    for _ in range(sub):
        cache = {}; nf = []; vl = list(v)
        def mid(a, b):
            k = (min(a, b), max(a, b))
            if k not in cache:
                m = (vl[a] + vl[b]) / 2; vl.append(m / np.linalg.norm(m)); cache[k] = len(vl) - 1
            return cache[k]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [[a, ab, ca], [b, bc, ab], [c, ca, bc], [ab, bc, ca]]
        v, f = np.array(vl), np.array(nf)
    return v, f
sv, sf = icosphere(3)   # 1280 tris
print(sv.shape, sf.shape)
rng = np.random.default_rng(0)
d = {"type": "scene", "cam": {"type": "perspective", "fov": 50, "to_world": look_at([0, -6, 4], [0, 0, 0.5], [0, 0, 1]),
     "film": {"type": "hdrfilm", "width": 512, "height": 512, "rfilter": {"type": "gaussian"}}, "sampler": {"type": "independent", "sample_count": 16}}}
fv = np.array([[-6, -6, 0], [6, -6, 0], [6, 6, 0], [-6, 6, 0]], float); ff = np.array([[0, 1, 2], [0, 2, 3]])
d["floor"] = {"type": "mesh", "vertices": fv, "faces": ff, "face_normals": True, "bsdf": {"type": "diffuse"}}
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for i in range(n):
    c = np.array([rng.uniform(-4, 4), rng.uniform(-4, 4), rng.uniform(0.3, 2.0)]); r = rng.uniform(0.15, 0.35)
    d[f"s{i}"] = {"type": "mesh", "vertices": sv * r + c, "faces": sf,
                  "bsdf": {"type": "roughconductor", "material": "Al", "distribution": "ggx", "alpha": 0.05} if i % 3 == 0 else {"type": "diffuse"}}
lv = np.array([[-1, -1, 6], [1, -1, 6], [1, 1, 6], [-1, 1, 6]], float)
d["light"] = {"type": "mesh", "vertices": lv, "faces": ff[:, ::-1], "face_normals": True, "emitter": {"type": "area", "radiance": {"type": "rgb", "value": 20.0}}}
dev = "cuda" if torch.cuda.is_available() else "cpu"
t = time.time(); sc = Scene.from_dict(d, device=dev); print("scene build s:", time.time() - t, "tris", sc.n_triangles if hasattr(sc, "n_triangles") else "?")
if dev == "cuda":
    def timed(fn, n=5):
        fn(); fn(); out = []
        for _ in range(n):
            torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); out.append((time.perf_counter() - t) * 1e3)
        return sorted(out)[n // 2]
    N = 512 * 512 * 16
    sc.tile_paths = int(sys.argv[2]) if len(sys.argv) > 2 else sc.tile_paths
    for mode in ("mega", "wavefront"):
        sc.tracer = mode
        ms = timed(lambda: sc.render_primal(sensor=0, seed=0, spp=16, max_depth=4))
        print(f"[{mode}] primal 512x512@16 ({N} paths, tiles of {sc.tile_paths}): {ms:.2f} ms = {N/ms/1e3:.1f} Mpaths/s")
        ms = timed(lambda: sc.trace_paths(sensor=0, seed=0, spp=16, max_depth=4))
        print(f"[{mode}] trace+log: {ms:.2f} ms = {N/ms/1e3:.1f} Mpaths/s")
        ms = timed(lambda: sc.trace_paths(sensor=0, seed=0, spp=16, max_depth=4, sparse_log=True))
        print(f"[{mode}] trace+sparse log: {ms:.2f} ms = {N/ms/1e3:.1f} Mpaths/s")
