"""Rays per bounce of the 128 004-triangle scene's gradient image (512x512 @ 64 spp, max_depth 4), from the packed log's
flag words: paths with vertex k logged (= closest-hit rays that found something at bounce k-1) and with an emitter sample
at vertex k (= visibility rays of bounce k-1)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from epsm_mitsuba3_amd.exp import clutter
dev = torch.device("cuda", 0)
scene = clutter.load_scene(dev, n_spheres=100, res=512, spp=64)
scene.tracer = "wavefront"
act = [0] * 6; em = [0] * 6; n = 0
for tr in scene.iter_traces(sensor=2, seed=1, spp=64, max_depth=clutter.max_depth, sparse_log=True, packed_log=True):
    fl = tr.log.flags if hasattr(tr.log, "flags") else tr.log.t_flags
    fl = fl.to(torch.int64)
    n += fl.numel()
    for k in range(5):
        w = (fl >> (5 * k)) & 31
        act[k] += int(((w & 4) != 0).sum()); em[k] += int(((w & 8) != 0).sum())
print("paths", n)
print("vertex k active (EPSM_FLAG_ACTIVE):", act[:5])
print("emitter sample at vertex k (EPSM_FLAG_ACTIVE_EM):", em[:5])
