"""prb_reparam's geometry gradients against finite differences, the reference's recipe (test_ad_integrators.py:833-871)
on restatements of its reparam configs (tests/_reparam_scenes.py).

    python tools/try_reparam_fd.py [CONFIG ...] [--device cpu|cuda] [--spp N] [--seeds K] [--rays R] [--weights ones|ramp]

On the CPU the host build of the tracer is used (tests/host_harness)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np

from _reparam_scenes import CONFIGS, fd_check

ap = argparse.ArgumentParser()
ap.add_argument("configs", nargs="*", default=list(CONFIGS))
ap.add_argument("--device", default="cpu")
ap.add_argument("--spp", type=int, default=256)
ap.add_argument("--seeds", type=int, default=2)
ap.add_argument("--rays", type=int, default=64)
ap.add_argument("--fd-spp-mult", type=int, default=4)
ap.add_argument("--fd-eps", type=float, default=0.0)
ap.add_argument("--kappa", type=float, default=0.0)
ap.add_argument("--reparam-depth", type=int, default=-1)
ap.add_argument("--weights", default="ramp", help="ones: grad_in = 1 (the reference's test); ramp: 0.5 + x / width")
a = ap.parse_args()

for name in a.configs:
    got, fd, dt = fd_check(name, a.device, a.spp, a.seeds, a.rays, a.weights, a.fd_spp_mult, a.fd_eps, a.kappa, a.reparam_depth)
    gm, fm = float(np.mean(got)), float(np.mean(fd))
    print(f"{name:28s} grad {gm:+.4f} (per seed {[round(x, 3) for x in got]})  FD {fm:+.4f} (per seed {[round(x, 3) for x in fd]})  "
          f"rel err {abs(gm - fm) / max(abs(fm), 1e-3):.3f}  [{dt:.1f} s backward]", flush=True)
