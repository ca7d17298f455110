"""render_backward of the 128 004-triangle scene (exp/clutter.py), 512x512 @ 64 spp = 2^24 paths, with and without
EPSM_TRACE_FUSE_FIRST_HIT (integrator property fuse_first_hit): wall-clock per call, median of 7, both variants.
python tools/time_first_hit_fusion.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import epsm_mitsuba3_amd as epsm
from epsm_mitsuba3_amd.exp import clutter
dev = torch.device("cuda", 0)
res, spp = 512, 64
scene = clutter.load_scene(dev, n_spheres=100, res=res, spp=spp)
for i in range(0, 100, 3):
    scene.attach(f"s{i}", positions=True, normals=True)
g = torch.Generator(device=dev).manual_seed(2)
grad_in = torch.randn((res, res, 5), generator=g, device=dev) * 1e-3
for variant in ("manifold", "manifold_caustic"):
    ref = None
    for fuse in (False, True, False, True):
        integ = epsm.load_dict({"type": variant, "max_depth": clutter.max_depth, "fuse_first_hit": fuse})
        integ.backward_spp = spp
        params = scene.param_grads()
        integ.render_backward(scene, params, grad_in, seed=1)
        torch.cuda.synchronize()
        out = []
        for _ in range(7):
            params.zero_()
            torch.cuda.synchronize(); t = time.perf_counter()
            integ.render_backward(scene, params, grad_in, seed=1)
            torch.cuda.synchronize(); out.append((time.perf_counter() - t) * 1e3)
        flat = params.flat.double().cpu()
        if ref is None:
            ref = flat
        print(f"{variant}: fuse_first_hit={fuse}: {sorted(out)[3]:.3f} ms per gradient image of {res * res * spp} paths "
              f"(min {min(out):.3f}); max |diff| to the unfused buffers / max |buffer| = {float((flat - ref).abs().max() / ref.abs().max()):.2e}")
