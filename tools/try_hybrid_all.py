import sys
sys.path.insert(0, '/root/repo')
from epsm_mitsuba3_amd.optim import run
from epsm_mitsuba3_amd.exp import bathroom, plate, slab
import numpy as np
for name, mod, method, kw in (("bathroom", bathroom, "manifold_hybrid", dict(iterations=100, lr=0.02)),
                              ("plate", plate, "manifold_hybrid", dict(iterations=70, lr=None)),
                              ("slab", slab, "manifold_caustic_hybrid", dict(iterations=70, lr=None))):
    old = mod.thres
    mod.thres = 40
    try:
        kw = {k: v for k, v in kw.items() if v is not None}
        hist, opt = run(method, name, log=lambda s: None, **kw)
        print(name, method, "start", round(hist[0], 3), "at switch", round(hist[40], 4), "end (last-10 mean)", round(float(np.mean(hist[-10:])), 4),
              "history", [round(h, 3) for h in hist[36::4]], flush=True)
    except Exception as e:
        print(name, "FAILED", repr(e)[:300], flush=True)
    finally:
        mod.thres = old
