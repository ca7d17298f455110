import sys; sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch
from test_tracer_wavefront_host import _rich_scene, _all_arrays
dev = torch.device("cuda", 0)
res, spp = 48, 16
sc = _rich_scene(res, spp, point_light=True, occluder=True, device=dev)
n = res * res * spp
sc.tracer = "mega"; a = sc._trace(0, seed=5, spp=spp, max_depth=3, K=2, lo=0, hi=n)
sc.tracer = "wavefront"; b = sc._trace(0, seed=5, spp=spp, max_depth=3, K=2, lo=0, hi=n)
torch.cuda.synchronize()
x, y = _all_arrays(a), _all_arrays(b)
for name in x:
    u, v = x[name].reshape(n, -1), y[name].reshape(n, -1)
    if u.dtype == torch.float32:
        d = ~(torch.isclose(u, v, rtol=1e-4, atol=1e-5) | (torch.isnan(u) & torch.isnan(v))).all(dim=1)
    else:
        d = (u != v).any(dim=1)
    if int(d.sum()):
        idx = torch.nonzero(d).flatten()[:3]
        print(name, int(d.sum()), "e.g.", idx.tolist(), u[idx].tolist(), v[idx].tolist())
