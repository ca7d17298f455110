"""Real scene (exp/clutter.py, 128 004 triangles), 512x512 @ 64 spp, max_depth 4: trace + log and backward pass on the
per-field log and on the native packed log, per stage (median of 5, wall-clock with synchronisation)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import epsm_mitsuba3_amd as epsm
from epsm_mitsuba3_amd.exp import clutter

dev = torch.device("cuda", 0)
res, spp = 512, 64
scene = clutter.load_scene(dev, n_spheres=100, res=res, spp=spp)
for i in range(0, 100, 3):
    scene.attach(f"s{i}", positions=True, normals=True)
g = torch.Generator(device=dev).manual_seed(2)
grad_in = torch.randn((res, res, 5), generator=g, device=dev) * 1e-3


def timed(fn, n=5):
    fn(); out = []
    for _ in range(n):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize()
        out.append((time.perf_counter() - t) * 1e3)
    return sorted(out)[n // 2]


for tracer in ("wavefront", "mega"):
    scene.tracer = tracer
    for packed in (False, True):
        integ = epsm.load_dict({"type": "manifold", "max_depth": clutter.max_depth, "packed_log": packed})
        integ.backward_spp = spp
        params = scene.param_grads()
        kw = dict(sensor=2, seed=1, spp=spp, max_depth=clutter.max_depth, sparse_log=True)

        def trace_only():
            for tr in scene.iter_traces(packed_log=packed, **kw):
                del tr
        tiles = list(scene.iter_traces(packed_log=packed, **kw))

        def backward_only():
            for tr in tiles:
                integ.backward_from_trace(tr, params, grad_in)
        t_total = timed(lambda: integ.render_backward(scene, params, grad_in, seed=1))
        t_trace = timed(trace_only)
        t_back = timed(backward_only)
        print(f"{tracer:9s} packed={packed!s:5s} render_backward {t_total:7.2f} ms   trace+log {t_trace:7.2f} ms   backward on resident tiles {t_back:6.2f} ms   ({len(tiles)} tiles)", flush=True)
        del tiles
