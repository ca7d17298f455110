"""What a ray costs the memory system: BVH nodes fetched (128 B each) and triangles tested (36 B each) per ray, by kind, on
the clutter scene -- counted by the host build of the traversal (tests/host_harness compiled with -DEPSM_TRAV_STATS).
    python tools/count_traversal.py [N_SPHERES] [RES] [SPP]"""
import ctypes as C
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from epsm_mitsuba3_amd.exp import clutter
from epsm_mitsuba3_amd.scene import Scene

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
res = int(sys.argv[2]) if len(sys.argv) > 2 else 128
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 4
here = os.path.dirname(os.path.abspath(__file__))
src = os.path.join(here, "..", "tests", "host_harness", "trace_host.cpp")
so = "/tmp/libtrace_host_stats.so"
stats_c = "/tmp/trav_stats.cpp"
open(stats_c, "w").write('extern "C" { long long g_trav_nodes[2] = {0, 0}, g_trav_tris[2] = {0, 0}, g_trav_rays[2] = {0, 0}; }\n')
subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fopenmp", "-Wno-unknown-pragmas", "-ffp-contract=off", "-DEPSM_TRAV_STATS",
                "-o", so, src, stats_c], check=True)
lib = C.CDLL(so)
for nm in ("epsm_trace_paths", "epsm_film_splat", "epsm_film_develop"):
    getattr(lib, nm).restype = C.c_int
sc = Scene.from_dict(clutter.scene_dict(n, res, spp), device="cpu")
sc._backend = lib
sc.tracer = "mega"
arr = lambda name: (C.c_longlong * 2).in_dll(lib, name)
sc.render_primal(sensor=0, seed=0, spp=spp, max_depth=4)
print(f"{sc.c_scene.n_triangles} triangles, {sc.c_scene.n_nodes} four-wide nodes ({sc.c_scene.n_nodes * 128 / 1e6:.2f} MB) + {sc.c_scene.n_triangles * 36 / 1e6:.2f} MB of "
      f"leaf triangles; {res}x{res} @ {spp} spp, max_depth 4")
for k, what in ((0, "closest hit (primary + bounce rays)"), (1, "any hit (visibility rays)")):
    r, nd, tr = arr("g_trav_rays")[k], arr("g_trav_nodes")[k], arr("g_trav_tris")[k]
    print(f"{what}: {r} rays, {nd / r:.1f} nodes + {tr / r:.1f} triangles per ray = {(128 * nd + 36 * tr) / r:.0f} B per ray")
