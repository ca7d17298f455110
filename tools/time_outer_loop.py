"""Wall-clock per iteration of the outer loop (epsm_mitsuba3_amd/optim.py) for an experiment module: render -> matcher ->
render_backward -> chain rule -> Adam.   python tools/time_outer_loop.py EXP [ITERATIONS] [key=value ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
import torch
from epsm_mitsuba3_amd import optim
exp = sys.argv[1] if len(sys.argv) > 1 else "bathroom"
its = int(sys.argv[2]) if len(sys.argv) > 2 else 20
tasks = importlib.import_module(f"epsm_mitsuba3_amd.exp.{exp}")
for k, v in (a.split("=") for a in sys.argv[3:]):
    old = getattr(tasks, k)
    setattr(tasks, k, v if isinstance(old, str) else type(old)(float(v)))
stamps = []
def log(s):
    torch.cuda.synchronize(); stamps.append(time.perf_counter())
optim.run("manifold", exp, iterations=its, log=log)
d = [b - a for a, b in zip(stamps[2:-1], stamps[3:])]
print(f"{exp}: resolution {tasks.resolution}, spp {tasks.spp}, match_res {tasks.match_res}, matcher {getattr(tasks, 'matcher', 'Sinkhorn')}: "
      f"{1e3 * sorted(d)[len(d) // 2]:.1f} ms per iteration (median of {len(d)})")
