"""Wall-clock of prb_reparam's render_backward (primal replay + film adjoints + the reparameterised pass) on the
128 k-triangle clutter scene, every tenth sphere attached:  python tools/bench_reparam.py [RES] [SPP] [RAYS] [MAX_DEPTH]"""
import sys
import time

sys.path.insert(0, ".")
import torch

import epsm_mitsuba3_amd as epsm
from epsm_mitsuba3_amd.exp import clutter
from epsm_mitsuba3_amd.scene import Scene

res = int(sys.argv[1]) if len(sys.argv) > 1 else 256
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rays = int(sys.argv[3]) if len(sys.argv) > 3 else 16
depth = int(sys.argv[4]) if len(sys.argv) > 4 else 3
d = clutter.scene_dict(100, res, spp)
d["sensor0"]["film"]["sample_border"] = True
scene = Scene.from_dict(d, device="cuda")
if len(sys.argv) > 5:
    scene.tile_paths = int(sys.argv[5])
for i in range(0, 100, 10):
    scene.attach(f"s{i}", positions=True, normals=True)
integ = epsm.load_dict({"type": "prb_reparam", "max_depth": depth, "reparam_rays": rays})
torch.manual_seed(0)
g = torch.randn((res, res, 3), device="cuda") * 1e-2
params = scene.param_grads()


def once():
    integ.render_backward(scene, params, g, sensor=0, seed=1, spp=spp)


# run-to-run: the same call twice into fresh buffers (float atomics: the order of the additions differs)
pa, pb = scene.param_grads(), scene.param_grads()
integ.render_backward(scene, pa, g, sensor=0, seed=1, spp=spp)
integ.render_backward(scene, pb, g, sensor=0, seed=1, spp=spp)
sa = torch.stack([pa.mesh_pos(f"s{i}").sum(0) for i in range(0, 100, 10)])
sb = torch.stack([pb.mesh_pos(f"s{i}").sum(0) for i in range(0, 100, 10)])
print(f"two runs: per-vertex max |a - b| / max |a| = {float((pa.pos - pb.pos).abs().max() / pa.pos.abs().max()):.2e}; per-mesh translation "
      f"gradients max |a - b| / max |a| = {float((sa - sb).abs().max() / sa.abs().max()):.2e}; max |per-vertex| {float(pa.pos.abs().max()):.3e}, "
      f"max |per-mesh sum| {float(sa.abs().max()):.3e}")
once(); once()
ts = []
for _ in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    once()
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
n = scene.sensors[0].wavefront_size(spp)
ms = sorted(ts)[1]
print(f"prb_reparam render_backward, {scene.c_scene.n_triangles} triangles, {res}x{res} (+border) @ {spp} spp = {n} paths, max_depth {depth}, "
      f"{rays} auxiliary rays per warp: {ms:.1f} ms = {n / ms / 1e3:.2f} M paths/s; |grad_pos| max {float(params.pos.abs().max()):.3e}")
