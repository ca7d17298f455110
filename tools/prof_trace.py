"""For rocprofv3: traces the big-scene wavefront `reps` times in one tracer mode.
    python tools/prof_trace.py MODE [n_spheres] [tile_paths] [reps]
With --diff instead of MODE: per-array count of paths on which the two tracer forms differ."""
import sys, runpy
sys.argv, args = [sys.argv[0], sys.argv[2] if len(sys.argv) > 2 else "100"], sys.argv
import torch
mode = args[1]
# reuse the scene of tools/bench_bigscene.py without its timing section
src = open(__file__.replace("prof_trace.py", "bench_bigscene.py")).read().split('if dev == "cuda":')[0]
ns = {}
exec(compile(src, "bench_bigscene_scene", "exec"), ns)
sc = ns["sc"]
sc.tile_paths = int(args[3]) if len(args) > 3 else 4194304
reps = int(args[4]) if len(args) > 4 else 3
if mode == "--diff":
    sys.path.insert(0, "tests")
    from test_tracer_wavefront_host import _all_arrays
    sc.tracer = "mega"; a = sc.trace_paths(sensor=0, seed=0, spp=16, max_depth=4)[0]
    sc.tracer = "wavefront"; b = sc.trace_paths(sensor=0, seed=0, spp=16, max_depth=4)[0]
    torch.cuda.synchronize()
    x, y = _all_arrays(a), _all_arrays(b)
    n = a.ray_o.shape[0]
    for name in x:
        u, v = x[name].reshape(n, -1), y[name].reshape(n, -1)
        if u.dtype == torch.float32:
            d = ~(torch.isclose(u, v, rtol=1e-4, atol=1e-5) | (torch.isnan(u) & torch.isnan(v))).all(dim=1)
        else:
            d = (u != v).any(dim=1)
        if int(d.sum()):
            print(f"{name}: {int(d.sum())} of {n} paths differ")
    print("diff done")
else:
    sc.tracer = mode
    for _ in range(reps):
        sc.trace_paths(sensor=0, seed=0, spp=16, max_depth=4)
    torch.cuda.synchronize()
    print("done", mode)
