"""render_backward of the 128 004-triangle scene at 512x512 @ 64 spp by wavefront tile size (median of 5, ms)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import epsm_mitsuba3_amd as epsm
from epsm_mitsuba3_amd.exp import clutter
from epsm_mitsuba3_amd.scene import Scene
dev = torch.device("cuda", 0)
res, spp = 512, 64
scene = clutter.load_scene(dev, n_spheres=100, res=res, spp=spp)
for i in range(0, 100, 3):
    scene.attach(f"s{i}", positions=True, normals=True)
scene.tracer = "wavefront"
g = torch.Generator(device=dev).manual_seed(2)
grad_in = torch.randn((res, res, 5), generator=g, device=dev) * 1e-3
integ = epsm.load_dict({"type": "manifold", "max_depth": clutter.max_depth})
integ.backward_spp = spp
params = scene.param_grads()
def timed(fn, n=5):
    fn(); out = []
    for _ in range(n):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize()
        out.append((time.perf_counter() - t) * 1e3)
    return sorted(out)[n // 2]
for lg in ((24,) if os.environ.get("TILES_ONLY_24") else (20, 21, 22, 23, 24)):
    Scene.WAVEFRONT_TILE_PATHS = 1 << lg
    scene.WAVEFRONT_TILE_PATHS = 1 << lg
    t = timed(lambda: integ.render_backward(scene, params, grad_in, seed=1))
    def trace_only():
        for tr in scene.iter_traces(sensor=2, seed=1, spp=spp, max_depth=clutter.max_depth, sparse_log=True, packed_log=True):
            del tr
    t2 = timed(trace_only)
    print(f"tile 2^{lg}: render_backward {t:7.2f} ms   trace+log {t2:7.2f} ms   peak memory {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB", flush=True)
