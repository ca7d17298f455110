"""exp/human.py on the GPU: error history of the outer loop (mean vertex distance to the target, metres)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from epsm_mitsuba3_amd import optim
from epsm_mitsuba3_amd.exp import human
its = int(sys.argv[1]) if len(sys.argv) > 1 else 60
lr = float(sys.argv[2]) if len(sys.argv) > 2 else None
for k, v in (a.split("=") for a in sys.argv[3:]):
    setattr(human, k, type(getattr(human, k))(float(v)) if not isinstance(getattr(human, k), str) else v)
opt_name = os.environ.get("OPT", "adam")
if opt_name == "sgd":
    import torch.optim as _o
    _Adam = _o.Adam
    _o.Adam = lambda params, lr: _o.SGD(params, lr=lr * float(os.environ.get("SGD_SCALE", "2.5")))      # OPT=sgd: plain gradient steps
t = time.time()
poses = []
def log(s):
    pass
extra = []
ot = []
_orig = human.optim_settings
def wrapped(scene):
    opt, a, b, out = _orig(scene)
    gt = human.gt_scene(scene.device).render_primal(sensor=0, seed=777, spp=256, max_depth=human.max_depth)
    def out2(o):
        e = out(o)
        a(scene, o)
        img = scene.render_primal(sensor=0, seed=778, spp=256, max_depth=human.max_depth)
        extra.append(float(((img[..., :3] - gt[..., :3]) ** 2).mean()))
        from epsm_mitsuba3_amd.matcher import Matcher, sinkhorn_divergence_and_grad_hip
        m = Matcher(human.match_res, scene.device)
        lo = lambda t: optim.resize(optim.to_ldr(t[..., :3]), human.match_res).reshape(-1, 3)
        ca = torch.cat([lo(img).clamp(0, 1), m.pos], 1); cb = torch.cat([lo(gt).clamp(0, 1), m.pos], 1)
        ot.append(float(sinkhorn_divergence_and_grad_hip(ca, cb)[0]))
        return e
    return opt, a, b, out2
human.optim_settings = wrapped
hist, opt = optim.run("manifold", "human", iterations=its, lr=lr, log=log)
print("seconds", round(time.time() - t, 1))
print("history", [round(h, 4) for h in hist])
print("image mse x1e4", [round(e * 1e4, 3) for e in extra])
print("sinkhorn divergence x1e6 of the 256-spp render from the target", [round(e * 1e6, 2) for e in ot])
tp = human.target_pose().reshape(24, 3)
p = opt["pose"].detach().cpu().reshape(24, 3)
print("pose error", float((p - tp).norm()), "of", float(tp.norm()))
for j in range(24):
    print(j, [round(float(x), 3) for x in p[j]], "target", [round(float(x), 3) for x in tp[j]])
