"""How full are the groups of the warp kernel's launch?  Reads the per-path request counts stage 1 left in the workspace."""
import sys
sys.path.insert(0, ".")
import torch
import epsm_mitsuba3_amd as epsm
from epsm_mitsuba3_amd.exp import clutter
from epsm_mitsuba3_amd.scene import Scene
res, spp, rays, depth = 512, 16, 16, 3
d = clutter.scene_dict(100, res, spp)
d["sensor0"]["film"]["sample_border"] = True
scene = Scene.from_dict(d, device="cuda")
for i in range(0, 100, 10):
    scene.attach(f"s{i}", positions=True, normals=True)
integ = epsm.load_dict({"type": "prb_reparam", "max_depth": depth, "reparam_rays": rays})
g = torch.randn((res, res, 3), device="cuda") * 1e-2
params = scene.param_grads()
integ.render_backward(scene, params, g, sensor=0, seed=1, spp=spp)
torch.cuda.synchronize()
N = scene.sensors[0].wavefront_size(spp)
ws = scene._reparam_ws
req_bytes = (N * 13 * 64 + 255) // 256 * 256
cnt = ws[req_bytes: req_bytes + 4 * N].view(torch.int32)
print("paths", N, "mean requests per path %.3f" % float(cnt.float().mean()), "hist", torch.bincount(cnt.long()).tolist())
n_max = 1 + 2 * depth
tot_groups = live_groups = nonempty_waves = 0
for n in range(n_max):
    live = (cnt > n)
    pad = (-N) % 4
    l4 = torch.nn.functional.pad(live, (0, pad)).view(-1, 4).sum(1)
    tot_groups += int(live.numel()); live_groups += int(live.sum()); nonempty_waves += int((l4 > 0).sum())
    print(f"n={n}: live groups {float(live.float().mean()):.3f}; waves (4 groups) empty {float((l4 == 0).float().mean()):.3f} full {float((l4 == 4).float().mean()):.3f}")
print(f"launched groups {tot_groups}, live {live_groups} ({live_groups / tot_groups:.3f}); lanes busy in non-empty waves {live_groups / (4 * nonempty_waves):.3f}")
