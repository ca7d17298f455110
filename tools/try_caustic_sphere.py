import sys, os
sys.path.insert(0, os.getcwd())
from epsm_mitsuba3_amd.optim import run
for lr in (0.03, 0.02):
    hist, opt = run("manifold_caustic", "caustic_sphere", iterations=60, lr=lr, log=lambda s: None)
    print("lr", lr, "start %.3f" % hist[0], "min last15 %.3f" % min(hist[-15:]), "last %.3f" % hist[-1], [round(h, 2) for h in hist[::6]], flush=True)
