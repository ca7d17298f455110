"""exp/human.py, the deciding experiment of VERDICT r3 item 3: does EPSM's pose pseudo-gradient agree with FINITE DIFFERENCES of the
matcher's OWN loss?  At three poses of the reference-settings loop -- the zero pose it starts from, the minimum around iteration
30, the plateau -- the Sinkhorn divergence between the rendered cloud and the target cloud (the value whose gradient field
``Matcher.match_Sinkhorn`` hands to ``render_backward``) is differentiated by central differences along the ~12 pose angles the
image determines best (largest diagonal of J = d(pose gradient)/d(pose)), under common random numbers (one fixed seed, FD_SPP
samples per pixel, a clean target), and compared with the seed-averaged EPSM pose gradient: cosine and per-angle sign, for the
whole gradient, for the first-hit term alone (si_follow.p * diffuse_grad[0], epsm.py:561-562) and for the occluder term alone
(the shadow, epsm.py:609-620).

    python tools/try_human_fd.py [FD_SPP=2048] [H=0.02] [SEEDS=4] [N_ANGLES=12]

The loop's images are tone-mapped and rounded to 8 bits before the matcher (optim.py:121,131); the finite differences use the
same tone map WITHOUT the rounding (a staircase has no derivative)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from epsm_mitsuba3_amd import integrators, load_dict
from epsm_mitsuba3_amd.exp import human as tasks
from epsm_mitsuba3_amd.integrators import render_seeds
from epsm_mitsuba3_amd.matcher import Matcher, sinkhorn_divergence_and_grad_hip
from epsm_mitsuba3_amd.optim import resize, to_ldr
from epsm_mitsuba3_amd.records import PackedRecords, PackedScatter
from epsm_mitsuba3_amd.tangent_scatter import first_vertex_tangent, manifold_grad_scatter

dev = "cuda"
FD_SPP = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
H = float(sys.argv[2]) if len(sys.argv) > 2 else 0.02
S = int(sys.argv[3]) if len(sys.argv) > 3 else 4
NA = int(sys.argv[4]) if len(sys.argv) > 4 else 12
IT_MIN, IT_PLATEAU = 30, 150


def smooth_ldr(img):
    x = img.clamp(0, 1)
    return torch.where(x <= 0.0031308, 12.92 * x, 1.055 * x.clamp_min(1e-12).pow(1 / 2.4) - 0.055).clamp(0, 1)


class Probe(integrators.ManifoldIntegrator):
    """The manifold integrator with one of its two terms switched off (per-field records, calc_grad + scatter in one launch)."""
    body = (0, 0)
    mode = "ref"

    def backward_from_trace(self, trace, params, grad_in, packed=None, out=None, mark=None, fused=None):
        d_ = trace.ray_d.device
        rec, sc = PackedRecords(trace.path_info, device=d_), PackedScatter(trace.scatter_info, device=d_)
        first = trace.path_info[1]
        dlduv, dldp, _ = first_vertex_tangent(trace.ray_o, trace.ray_d, trace.ray_dx, trace.ray_dy, grad_in, trace.spp, trace.res,
                                              first["points"][0], first["points"][1], first["points"][2], first["active"],
                                              dlduv_width=2, want_origin_grad=True, path_offset=trace.path_offset)
        tri = trace.scatter_info[0]["tri"].long()
        on_body = (tri >= self.body[0]) & (tri < self.body[1])
        if self.mode == "shadow":                    # occluder term alone: the figure's own first hits give nothing
            dldp = torch.where(on_body[:, None], torch.zeros_like(dldp), dldp)
        if self.mode == "first" and sc.packed[0].get("shadow") is not None:      # first-hit term alone: no occluder record
            sc.packed[0]["shadow"][:, 0] = -1
        manifold_grad_scatter(self.variant, rec, sc, dlduv, dldp.contiguous(), params.pos, params.nrm,
                              params.alpha if params.B else None, clip=self.outlier_clip)


scene = tasks.load_scene(dev)
ref_integ = load_dict({"type": "manifold", "max_depth": tasks.max_depth})
probe = Probe({"max_depth": tasks.max_depth, "packed_log": False})
Probe.body = scene.mesh_tri_slices["human"]
gts = tasks.gt_scene(dev)
gt_loop = resize(to_ldr(gts.render_primal(sensor=0, seed=0, spp=512, max_depth=tasks.max_depth)), tasks.match_res)       # what the loop matches against
gt_clean = resize(smooth_ldr(gts.render_primal(sensor=0, seed=0, spp=8192, max_depth=tasks.max_depth)[..., :3]), tasks.match_res)
matcher = Matcher(tasks.match_res, dev)
opt, apply_t, backward, output = tasks.optim_settings(scene)
rep = tasks.resolution // tasks.match_res
target_cloud = torch.cat([gt_clean.reshape(-1, 3).clamp(0, 1), matcher.pos], dim=1)


def set_pose(pose):
    with torch.no_grad():
        opt["pose"].copy_(pose)
    apply_t(scene, opt)


def loss_at(pose, seed=12345):
    """The matcher's own loss: Sinkhorn divergence of the rendered cloud from the (clean) target cloud, common random numbers."""
    set_pose(pose)
    img = scene.render_primal(sensor=1, seed=seed, spp=FD_SPP, max_depth=tasks.max_depth)[..., :3]
    low = resize(smooth_ldr(img), tasks.match_res)
    cloud = torch.cat([low.reshape(-1, 3).clamp(0, 1), matcher.pos], dim=1)
    loss, _ = sinkhorn_divergence_and_grad_hip(cloud, target_cloud, matcher.blur, matcher.scaling)
    return float(loss) * tasks.match_res ** 2                                     # matcher.py:60 scales the gradient by res^2


def epsm_grad(pose, integ, seeds, gt_low):
    G = torch.zeros(72, device=dev)
    for seed in seeds:
        set_pose(pose)
        s, sg = render_seeds(1000 + seed)
        img = integ.render(scene, sensor=1, seed=s, spp=tasks.spp)
        params = scene.param_grads()
        low = resize(to_ldr(img[..., :3]), tasks.match_res)
        g = matcher.match_Sinkhorn(low.reshape(-1, 3), gt_low.reshape(-1, 3)).reshape(tasks.match_res, tasks.match_res, 5).repeat(rep, rep, 1)
        integ.render_backward(scene, params, g, sensor=1, seed=sg, spp=tasks.spp)
        backward(opt, params)
        G += torch.nan_to_num(opt["pose"].grad.detach().reshape(-1))
    return G / len(seeds)


# ---- the loop at the reference's settings, poses kept at the start, the minimum and the plateau
poses = {"zero pose (iteration 0)": torch.zeros(1, 72, device=dev)}
optimizer = torch.optim.Adam([opt["pose"]], lr=tasks.lr)
hist = [output(opt)]
for it in range(IT_PLATEAU):
    apply_t(scene, opt)
    s, sg = render_seeds(it)
    img = ref_integ.render(scene, sensor=1, seed=s, spp=tasks.spp)
    params = scene.param_grads()
    low = resize(to_ldr(img[..., :3]), tasks.match_res)
    g = matcher.match_Sinkhorn(low.reshape(-1, 3), gt_loop.reshape(-1, 3)).reshape(tasks.match_res, tasks.match_res, 5).repeat(rep, rep, 1)
    ref_integ.render_backward(scene, params, g, sensor=1, seed=sg, spp=tasks.spp)
    backward(opt, params)
    opt["pose"].grad = torch.nan_to_num(opt["pose"].grad, nan=0.0, posinf=0.0, neginf=0.0)
    optimizer.step()
    hist.append(output(opt))
    if it + 1 == IT_MIN:
        poses[f"minimum (iteration {IT_MIN})"] = opt["pose"].detach().clamp(-tasks.POSE_CLAMP, tasks.POSE_CLAMP).clone()
poses[f"plateau (iteration {IT_PLATEAU})"] = opt["pose"].detach().clamp(-tasks.POSE_CLAMP, tasks.POSE_CLAMP).clone()
print(f"# exp/human.py at the reference's settings, primal / differential seeds de-correlated: vertex distance {hist[0] * 100:.2f} cm at the start, "
      f"{min(hist) * 100:.2f} cm at its minimum (iteration {hist.index(min(hist))}), {hist[IT_MIN] * 100:.2f} cm at {IT_MIN}, "
      f"{sum(hist[-20:]) / 20 * 100:.2f} cm over the last 20 of {IT_PLATEAU}")

# ---- the angles the image determines best: largest diagonal of J at the zero pose (one seed, central differences of the field)
zero = poses["zero pose (iteration 0)"]
if os.environ.get("HUMAN_FD_ANGLES"):                # (a previous run's list: skips the 144 gradient evaluations)
    angles = [int(a) for a in os.environ["HUMAN_FD_ANGLES"].split(",")][:NA]
    print(f"# angles (joint, axis) taken from HUMAN_FD_ANGLES: {[(a // 3, a % 3) for a in angles]}")
else:
    diag = torch.zeros(72)
    for j in range(72):
        e = torch.zeros(1, 72, device=dev); e[0, j] = 0.03
        diag[j] = float((epsm_grad(zero + e, ref_integ, [0], gt_loop)[j] - epsm_grad(zero - e, ref_integ, [0], gt_loop)[j]) / 0.06)
    angles = diag.argsort(descending=True)[:NA].tolist()
    print(f"# angles (joint, axis) with the largest diagonal of J at the zero pose: {[(a // 3, a % 3) for a in angles]}; "
          f"J_jj from {float(diag[angles[0]]):.2f} down to {float(diag[angles[-1]]):.2f} (median of all 72: {float(diag.median()):.3f})")
tp = tasks.target_pose().to(dev)
print(f"# target pose at those angles: {[round(float(tp[0, a]), 3) for a in angles]}")

cos = lambda a, b: float((a * b).sum() / (a.norm() * b.norm()).clamp_min(1e-30))
for name, pose in poses.items():
    L0 = loss_at(pose)
    fd = torch.zeros(NA)
    for n, a in enumerate(angles):
        e = torch.zeros(1, 72, device=dev); e[0, a] = H
        fd[n] = (loss_at(pose + e) - loss_at(pose - e)) / (2 * H)
    # noise floor of the differences themselves: the same difference under another seed
    fd2 = torch.zeros(NA)
    for n, a in enumerate(angles[:4]):
        e = torch.zeros(1, 72, device=dev); e[0, a] = H
        fd2[n] = (loss_at(pose + e, seed=777) - loss_at(pose - e, seed=777)) / (2 * H)
    set_pose(pose)
    dist = output(opt)
    at_clamp = int((pose.abs() >= tasks.POSE_CLAMP - 1e-6).sum())
    print(f"\n## {name}: vertex distance {dist * 100:.2f} cm, matcher loss x res^2 = {L0:.4f}, {at_clamp} of 72 angles at the clamp")
    print(f"   pose at those angles (clamp +-{tasks.POSE_CLAMP}):      {[round(float(pose[0, a]), 3) for a in angles]}")
    print(f"   FD of the loss (h = {H}, {FD_SPP} spp):        {[round(float(v), 3) for v in fd]}")
    # a clamped angle whose loss gradient points OUT of the box is held by the clamp, not by the field: leave those out
    free = torch.tensor([not ((float(pose[0, a]) >= tasks.POSE_CLAMP - 1e-6 and float(fd[n]) < 0) or
                              (float(pose[0, a]) <= -tasks.POSE_CLAMP + 1e-6 and float(fd[n]) > 0)) for n, a in enumerate(angles)])
    print(f"   angles NOT held by the clamp: {int(free.sum())} of {NA}")
    print(f"   (first four under another seed:               {[round(float(v), 3) for v in fd2[:4]]})")
    for mode in ("ref", "first", "shadow"):
        Probe.mode = mode
        g = epsm_grad(pose, ref_integ if mode == "ref" else probe, list(range(S)), gt_loop).cpu()[angles]
        signs = int(((g * fd) > 0).sum())
        big = fd.abs() > 0.25 * fd.abs().max()
        print(f"   EPSM pose gradient, {mode:6s} ({S} seeds):      {[round(float(v), 3) for v in g]}")
        print(f"      free angles only: cosine {cos(g[free], fd[free]):+.3f}, same sign on {int(((g * fd) > 0)[free].sum())} of {int(free.sum())}")
        print(f"      cosine with FD {cos(g, fd):+.3f}; same sign on {signs} of {NA} angles ({int(((g * fd) > 0)[big].sum())} of the {int(big.sum())} "
              f"with |FD| > a quarter of the largest); |g| / |FD| = {float(g.norm() / fd.norm().clamp_min(1e-30)):.3f}")
