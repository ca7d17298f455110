"""The backward pass on a TRACED tile (clutter scene, 2^24 paths, gradient-only native log), five launches -- for rocprofv3:
python tools/prof_real_backward.py [variant] [interleaved | dense]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import epsm_mitsuba3_amd as epsm
from epsm_mitsuba3_amd.exp import clutter
import bench

variant = sys.argv[1] if len(sys.argv) > 1 else "manifold"
layout = sys.argv[2] if len(sys.argv) > 2 else "interleaved"
dev = torch.device("cuda", 0)
res, spp = 512, 64
scene = clutter.load_scene(dev, n_spheres=100, res=res, spp=spp)
for i in range(0, 100, 3):
    scene.attach(f"s{i}", positions=True, normals=True)
scene.log_layout = layout
integ = epsm.load_dict({"type": variant, "max_depth": clutter.max_depth})
integ.backward_spp = spp
params = scene.param_grads()
g = torch.Generator(device=dev).manual_seed(2)
grad_in = torch.randn((res, res, 5), generator=g, device=dev) * 1e-3
kw = dict(sensor=2, seed=1, spp=spp, max_depth=clutter.max_depth, sparse_log=True, packed_log=True, gradient_only=variant)
tiles = list(scene.iter_traces(**kw))
for rep in range(3):
    del tiles
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    tiles = list(scene.iter_traces(**kw))
    e1.record(); torch.cuda.synchronize()
    print(f"[{layout}] trace + log {e0.elapsed_time(e1):.3f} ms")
log = tiles[0].log
live = bench.live_vertices(log.flags, log.K, variant)
n = log.flags.numel()
print(f"{variant}: {n} paths, live vertices per path {float(live.float().mean()):.3f}, paths with 0 / 1 / 2 / 3+ live vertices "
      f"{[int((live == k).sum()) for k in range(3)] + [int((live >= 3).sum())]}")
for rep in range(5):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for tr in tiles:
        integ.backward_from_trace(tr, params, grad_in)
    e1.record(); torch.cuda.synchronize()
    print(f"[{layout}] backward pass {e0.elapsed_time(e1):.3f} ms")
