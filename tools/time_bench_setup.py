import sys, time, os
sys.path.insert(0, os.getcwd())
import torch
import epsm_mitsuba3_amd as epsm
from epsm_mitsuba3_amd.records import PackedLog
dev = torch.device("cuda", 0)
res, spp, K, V = 1024, 256, 5, 100000
n = 1 << 24
scene = epsm.SyntheticScene(res=res, n_vertices=K, n_scene_vertices=V, n_bsdfs=4, profile="bathroom", device=dev, tile_paths=n)
def T(): torch.cuda.synchronize(); return time.perf_counter()
for s in range(3):
    t0 = T()
    trace = scene.tile(s, s * n, (s + 1) * n, seed=0, spp=spp, K=K, lean=True)
    t1 = T()
    log = PackedLog.from_trace(trace, device=dev, table=scene.triangle_table(), free=True)
    t2 = T()
    del trace
    torch.cuda.empty_cache()
    t3 = T()
    print(f"slab {s}: tile {t1-t0:.2f} s, pack {t2-t1:.2f} s, empty_cache {t3-t2:.2f} s", flush=True)
