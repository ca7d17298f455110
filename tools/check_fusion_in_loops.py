"""The outer loop with the wavefront tracer forced -- so that render_backward takes the fused route (EPSM_TRACE_FUSE_FIRST_HIT + the
survivors' list) on the small experiment scenes, which the one-launch tracer serves by default -- against the default route:
the same experiments end where they end without it.   python tools/check_fusion_in_loops.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import epsm_mitsuba3_amd.scene as S
from epsm_mitsuba3_amd.optim import run

orig = S.Scene._upload
for exp, method in (("slab", "manifold"), ("bathroom", "manifold"), ("caustic_sphere", "manifold_caustic"), ("camera", "manifold")):
    out = {}
    for forced in (False, True):
        def patched(self, *a, _f=forced, **k):
            r = orig(self, *a, **k)
            if _f:
                self.tracer = "wavefront"
            return r
        S.Scene._upload = patched
        hist, _ = run(method, exp, iterations=40, log=lambda s: None)
        out[forced] = (hist[0], min(hist), hist[-1])
    S.Scene._upload = orig
    print(f"{exp} / {method}: start, best, last  default tracer {tuple(round(v, 4) for v in out[False])}   wavefront + fusion {tuple(round(v, 4) for v in out[True])}", flush=True)
