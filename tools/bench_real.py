"""Wall-clock of the real pipeline on the analytic 'plate' scene: primal render and one backward pass
(trace with vertex log -> tangent -> fused gradient/scatter), per stage, via HIP events."""
import importlib, sys, time; sys.path.insert(0, ".")
import torch
import epsm_mitsuba3_amd as epsm

tasks = importlib.import_module("epsm_mitsuba3_amd.exp.plate")
dev = torch.device("cuda", 0)
res, spp = int(sys.argv[1]) if len(sys.argv) > 1 else 512, int(sys.argv[2]) if len(sys.argv) > 2 else 64
tasks.resolution, tasks.spp, tasks.match_res = res, spp, res // 2
scene = tasks.load_scene(dev)
scene.attach("light", positions=True); scene.attach("plate", positions=True)
integ = epsm.load_dict({"type": "manifold", "max_depth": 4})
def timed(fn, n=5):
    # median of n wall-clock repetitions (the first repetitions may still grow the caching allocator's pool)
    fn(); fn(); out = []
    for _ in range(n):
        torch.cuda.synchronize(); t = time.perf_counter()
        fn()
        torch.cuda.synchronize(); out.append((time.perf_counter() - t) * 1e3)
    return sorted(out)[n // 2]
N = res * res * spp
ms = timed(lambda: scene.render_primal(sensor=1, seed=0, spp=spp, max_depth=4))
print(f"primal render {res}x{res}@{spp}: {ms:.2f} ms  ({N/ms/1e3:.1f} Mpaths/s)")
integ.backward_spp = spp
s2 = scene.sensors[2]; s2.width = s2.height = res
grad_in = torch.randn((res, res, 5), device=dev) * 1e-3
params = scene.param_grads()
ms_t = timed(lambda: scene.trace_paths(sensor=2, seed=0, spp=spp, max_depth=4))
traces = scene.trace_paths(sensor=2, seed=0, spp=spp, max_depth=4)
ms_b = timed(lambda: [integ.backward_from_trace(t, params, grad_in) for t in traces])
print(f"backward {res}x{res}@{spp} = {N} paths: trace+log {ms_t:.2f} ms ({N/ms_t/1e3:.1f} Mpaths/s), tangent+grad+scatter {ms_b:.2f} ms, "
      f"total {ms_t+ms_b:.2f} ms")
