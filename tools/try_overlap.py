"""Does the trace of tile t+1 hide under the backward pass of tile t?  Clutter scene, 512 x 512 @ 256 spp = 2^26 paths = four
tiles of 2^24: (a) one stream: trace, backward pass, trace, ...; (b) two streams: the traces on one, the backward passes on the
other, each waiting for its tile's event.  python tools/try_overlap.py [variant] [spp] [tile_log2]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import epsm_mitsuba3_amd as epsm
from epsm_mitsuba3_amd.exp import clutter

variant = sys.argv[1] if len(sys.argv) > 1 else "manifold"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 256
tile_log2 = int(sys.argv[3]) if len(sys.argv) > 3 else 24
dev = torch.device("cuda", 0)
res = 512
scene = clutter.load_scene(dev, n_spheres=100, res=res, spp=spp)
scene.WAVEFRONT_TILE_PATHS = 1 << tile_log2
for i in range(0, 100, 3):
    scene.attach(f"s{i}", positions=True, normals=True)
integ = epsm.load_dict({"type": variant, "max_depth": clutter.max_depth})
integ.backward_spp = spp
params = scene.param_grads()
g = torch.Generator(device=dev).manual_seed(2)
grad_in = torch.randn((res, res, 5), generator=g, device=dev) * 1e-3
kw = dict(sensor=2, seed=1, spp=spp, max_depth=clutter.max_depth, sparse_log=True, packed_log=True, gradient_only=variant)
n = res * res * spp


def one_stream():
    for tr in scene.iter_traces(**kw):
        integ.backward_from_trace(tr, params, grad_in)
        del tr


s_tr, s_bw = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def two_streams():
    keep = []
    cur = torch.cuda.current_stream(dev)
    s_tr.wait_stream(cur); s_bw.wait_stream(cur)
    it = scene.iter_traces(**kw)
    while True:
        with torch.cuda.stream(s_tr):
            tr = next(it, None)
            if tr is None:
                break
            ev = torch.cuda.Event(); ev.record(s_tr)
        with torch.cuda.stream(s_bw):
            s_bw.wait_event(ev)
            integ.backward_from_trace(tr, params, grad_in)
        keep.append(tr)
    cur.wait_stream(s_tr); cur.wait_stream(s_bw)
    torch.cuda.synchronize()
    del keep


def timed(fn, reps=3):
    fn(); out = []
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize()
        out.append((time.perf_counter() - t) * 1e3)
    return sorted(out)[reps // 2]


params.flat.zero_(); one_stream(); torch.cuda.synchronize(); a = params.flat.clone()
params.flat.zero_(); two_streams(); torch.cuda.synchronize(); b = params.flat.clone()
m = a.abs().max().item()
print(f"{variant}, {n} paths in tiles of 2^{tile_log2}: gradients of the two forms differ by {(a - b).abs().max().item() / m:.2e} of the buffer")
t1 = timed(one_stream); t2 = timed(two_streams)
print(f"one stream {t1:.2f} ms = {n / t1 / 1e6:.2f} G paths/s; two streams {t2:.2f} ms = {n / t2 / 1e6:.2f} G paths/s", flush=True)
