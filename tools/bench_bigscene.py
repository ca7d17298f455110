import sys, time; sys.path.insert(0, ".")
import numpy as np, torch
from epsm_mitsuba3_amd.scene import Scene, look_at
from epsm_mitsuba3_amd.exp import clutter
import os
from epsm_mitsuba3_amd import scene as _S
if os.environ.get("EPSM_LEAF_SIZE") or os.environ.get("EPSM_SAH_MIN"):      # BVH build experiments
    _S.build_bvh.__defaults__ = (int(os.environ.get("EPSM_LEAF_SIZE", _S.LEAF_SIZE)), int(os.environ.get("EPSM_SAH_MIN", _S.SAH_MIN)))
    print("build_bvh defaults:", _S.build_bvh.__defaults__)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
d = clutter.scene_dict(n, 512, 16)
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
