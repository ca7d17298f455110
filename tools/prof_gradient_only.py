"""Per-bounce queue lengths and time of the backward trace with and without EPSM_TRACE_GRADIENT_ONLY on the clutter scene
(128 004 triangles, 512 x 512 @ 64 spp = 2^24 paths, packed log): python tools/prof_gradient_only.py [variant]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from epsm_mitsuba3_amd.exp import clutter

variant = sys.argv[1] if len(sys.argv) > 1 else "manifold"
dev = torch.device("cuda", 0)
res, spp = int(os.environ.get("EPSM_PROF_RES", 512)), int(os.environ.get("EPSM_PROF_SPP", 64))
sc = clutter.load_scene(dev, n_spheres=100, res=res, spp=spp)
sc.tracer = os.environ.get("EPSM_PROF_TRACER", "wavefront")
only = len(sys.argv) > 2 and sys.argv[2] == "only"           # (profiler runs: the gradient-only trace alone)
for go in ((variant,) if only else (None, variant)):
    times = []
    for rep in range(4):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        tr = list(sc.iter_traces(sensor=2, seed=3 + rep, spp=spp, max_depth=clutter.max_depth, packed_log=True, sparse_log=True, gradient_only=go))
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
        q = sc.wavefront_queue_lengths() if sc.use_wavefront() else {'alive': None, 'shadow': None}
        del tr
    print(f"gradient_only={go}: trace+log ms per {res * res * spp} paths {min(times[1:]):.2f} (runs {['%.2f' % t for t in times]}); alive into bounce b {q['alive']}; visibility rays {q['shadow']}", flush=True)
