import sys
sys.path.insert(0, '/root/repo')
from epsm_mitsuba3_amd.optim import run
from epsm_mitsuba3_amd.exp import bathroom
import numpy as np
for spp, res in ((16, 64), (256, 64), (64, 128)):
    bathroom.thres, bathroom.spp, bathroom.resolution = 40, spp, res
    hist, opt = run("manifold_hybrid", "bathroom", log=lambda s: None, iterations=100, lr=0.02)
    print("spp", spp, "res", res, "start", round(hist[0], 3), "at switch", round(hist[40], 4), "end (last-10 mean)", round(float(np.mean(hist[-10:])), 4),
          "history", [round(h, 3) for h in hist[36::4]], flush=True)
