"""For rocprofv3: `render_backward` of the 128 004-triangle scene (exp/clutter.py), 512x512 @ 64 spp, native packed log,
REPS times.   python tools/prof_render_backward.py [wavefront|mega] [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import epsm_mitsuba3_amd as epsm
from epsm_mitsuba3_amd.exp import clutter
dev = torch.device("cuda", 0)
res, spp = 512, 64
scene = clutter.load_scene(dev, n_spheres=100, res=res, spp=spp)
for i in range(0, 100, 3):
    scene.attach(f"s{i}", positions=True, normals=True)
scene.tracer = sys.argv[1] if len(sys.argv) > 1 else "wavefront"
g = torch.Generator(device=dev).manual_seed(2)
grad_in = torch.randn((res, res, 5), generator=g, device=dev) * 1e-3
integ = epsm.load_dict({"type": "manifold", "max_depth": clutter.max_depth})
integ.backward_spp = spp
params = scene.param_grads()
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
    integ.render_backward(scene, params, grad_in, seed=1)
torch.cuda.synchronize()
print("done")
