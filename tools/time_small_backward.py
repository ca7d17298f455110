"""Wall-clock of render_backward at the reference's OWN backward sizes (256 x 256 @ 8 spp = 524 288 paths: exp/human.py; 128 x 128
@ 8 spp = 131 072: exp/shadow.py): how much of it is GPU time.   python tools/time_small_backward.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import epsm_mitsuba3_amd as epsm
from epsm_mitsuba3_amd.exp import human, shadow, plate
dev = torch.device("cuda", 0)
for mod in (human, shadow, plate):
    sc = mod.load_scene(dev)
    mod.optim_settings(sc)
    res = sc.sensors[2].width
    integ = epsm.load_dict({"type": "manifold", "max_depth": mod.max_depth})
    g = torch.randn((res, res, 5), device=dev) * 1e-3
    p = sc.param_grads()
    for tracer in ("auto", "mega", "wavefront"):
        sc.tracer = tracer
        for _ in range(3):
            integ.render_backward(sc, p, g, seed=1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record()
        for _ in range(20):
            integ.render_backward(sc, p, g, seed=1)
        e1.record(); torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 20 * 1e3
        print(f"{mod.__name__.rsplit('.', 1)[-1]:8s} {res}x{res} @ 8 spp = {res * res * 8} paths, {sc.T} triangles, tracer {tracer:9s}: "
              f"render_backward {wall:.3f} ms wall, {e0.elapsed_time(e1) / 20:.3f} ms between events", flush=True)
