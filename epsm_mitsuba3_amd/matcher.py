"""Pixel matcher of the EPSM outer loop (EPSM/utils/matcher.py:10-63).

The reference calls ``geomloss.SamplesLoss("sinkhorn", blur=0.01, scaling=0.9)`` on two 5-D point
clouds (r,g,b,x,y) -- rendered pixels and target pixels on the same ``res x res`` grid -- and takes
the gradient w.r.t. the rendered points, times ``res^2`` (matcher.py:51-63).  geomloss / KeOps are
not installable here, so this is a plain-torch restatement of geomloss's *tensorized* debiased
Sinkhorn divergence (p = 2, cost |x-y|^2/2, uniform weights, epsilon-scaling from the point-cloud
diameter down to blur^2 by ``scaling^2`` per step, symmetric averaged updates, gradient through the
last extrapolation step with detached duals).  PARITY UNPINNED (no geomloss here to compare with);
pinned by properties in tests/test_matcher.py: zero gradient for identical clouds, gradient =
displacement for a translated cloud, descent reduces the divergence.

On the GPU the same iteration runs on ``epsm_sinkhorn_softmin`` (csrc/epsm_matcher.hip, include/epsm.h): the softmin of a
dual update evaluated online -- no n x m cost matrix, which is 17 GB at the 256 x 256 matching resolution of exp/human.py --
and the gradient of the last extrapolation in closed form (d softmin_i / d x_i = x_i - sum_j p_ij y_j).  Checked against
this file's dense torch form in tests/test_gpu_matcher.py.
"""
from __future__ import annotations

import math

import torch


def _softmin(eps: float, C: torch.Tensor, h: torch.Tensor) -> torch.Tensor:
    return -eps * torch.logsumexp(h[None, :] - C / eps, dim=1)


def _cost(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    return 0.5 * torch.cdist(x, y).pow(2)


def sinkhorn_divergence(x: torch.Tensor, y: torch.Tensor, blur: float = 0.01, scaling: float = 0.9) -> torch.Tensor:
    """Debiased Sinkhorn divergence S_eps(alpha, beta) with uniform weights; differentiable in ``x``."""
    n, m = x.shape[0], y.shape[0]
    a_log = torch.full((n,), -math.log(n), device=x.device, dtype=x.dtype)
    b_log = torch.full((m,), -math.log(m), device=x.device, dtype=x.dtype)
    with torch.no_grad():
        xd, yd = x.detach(), y.detach()
        mins = torch.minimum(xd.min(0).values, yd.min(0).values)
        maxs = torch.maximum(xd.max(0).values, yd.max(0).values)
        diameter = float((maxs - mins).norm().clamp_min(1e-12))
        eps_list = [diameter ** 2]
        e = 2 * math.log(diameter)
        while e > 2 * math.log(blur):
            eps_list.append(math.exp(e)); e += 2 * math.log(scaling)
        eps_list.append(blur ** 2)
        C_xy, C_yx, C_xx, C_yy = _cost(xd, yd), _cost(yd, xd), _cost(xd, xd), _cost(yd, yd)
        eps = eps_list[0]
        f_aa, g_bb = _softmin(eps, C_xx, a_log), _softmin(eps, C_yy, b_log)
        g_ab, f_ba = _softmin(eps, C_yx, a_log), _softmin(eps, C_xy, b_log)
        for eps in eps_list:
            ft_ba = _softmin(eps, C_xy, b_log + g_ab / eps)
            gt_ab = _softmin(eps, C_yx, a_log + f_ba / eps)
            f_ba, g_ab = 0.5 * (f_ba + ft_ba), 0.5 * (g_ab + gt_ab)
            f_aa = 0.5 * (f_aa + _softmin(eps, C_xx, a_log + f_aa / eps))
            g_bb = 0.5 * (g_bb + _softmin(eps, C_yy, b_log + g_bb / eps))
        eps = eps_list[-1]
    # last extrapolation with the graph attached to x (duals detached); both cross terms from the OLD duals, as geomloss's
    # sinkhorn_loop writes its last step (one simultaneous assignment)
    f_ba_new = _softmin(eps, _cost(x, yd), b_log + g_ab / eps)
    f_aa_new = _softmin(eps, _cost(x, xd), a_log + f_aa / eps)
    with torch.no_grad():
        g_ab_new = _softmin(eps, C_yx, a_log + f_ba / eps)
        g_bb_new = _softmin(eps, C_yy, b_log + g_bb / eps)
    return (f_ba_new - f_aa_new).mean() + (g_ab_new - g_bb_new).mean()


_scratch = {}


def softmin_hip(eps: float, x: torch.Tensor, y: torch.Tensor, h: torch.Tensor, want_wsum: bool = False):
    """``-eps * logsumexp_j(h_j - |x_i - y_j|^2 / (2 eps))`` for CUDA float32 clouds x (n,D), y (m,D), h (m): (n,) and,
    with ``want_wsum``, ``sum_j softmax_ij y_j`` (n,D).  Fails loudly without the HIP library."""
    import ctypes as C
    from . import _lib
    lib = _lib.lib()
    assert x.is_cuda and y.is_cuda and h.is_cuda, "softmin_hip takes device tensors"
    x, y, h = x.detach().contiguous().float(), y.detach().contiguous().float(), h.detach().contiguous().float()
    n, D = x.shape
    m = y.shape[0]
    assert y.shape[1] == D and h.shape == (m,)
    need = int(lib.epsm_sinkhorn_scratch_bytes(n, m, D))
    key = (x.device.index, need)
    sc = _scratch.get(key)
    if sc is None:
        _scratch.clear()
        sc = _scratch[key] = torch.empty((max(need, 4) + 3) // 4, dtype=torch.float32, device=x.device)
    out = torch.empty((n,), dtype=torch.float32, device=x.device)
    w = torch.empty((n, D), dtype=torch.float32, device=x.device) if want_wsum else None
    stream = torch.cuda.current_stream(x.device).cuda_stream
    _lib.check(lib.epsm_sinkhorn_softmin(n, m, D, x.data_ptr(), y.data_ptr(), h.data_ptr(), float(eps), out.data_ptr(),
                                         w.data_ptr() if want_wsum else None, sc.data_ptr(), sc.numel() * 4, C.c_void_p(stream)),
               "epsm_sinkhorn_softmin")
    return (out, w) if want_wsum else out


def update_hip(eps: float, x: torch.Tensor, y: torch.Tensor, dual, log_weight: float, prev=None, want_wsum: bool = False):
    """One dual update in one call (``epsm_sinkhorn_update``): softmin over y with h = log_weight + dual / eps, averaged with
    ``prev`` when given.  x (n,D), y (m,D), dual (m) or None, prev (n) or None: contiguous CUDA float32."""
    import ctypes as C
    from . import _lib
    lib = _lib.lib()
    n, D = x.shape
    m = y.shape[0]
    need = int(lib.epsm_sinkhorn_scratch_bytes(n, m, D))
    key = (x.device.index, need)
    sc = _scratch.get(key)
    if sc is None:
        if len(_scratch) > 8:
            _scratch.clear()
        sc = _scratch[key] = torch.empty((max(need, 4) + 3) // 4, dtype=torch.float32, device=x.device)
    out = torch.empty((n,), dtype=torch.float32, device=x.device)
    w = torch.empty((n, D), dtype=torch.float32, device=x.device) if want_wsum else None
    stream = torch.cuda.current_stream(x.device).cuda_stream
    _lib.check(lib.epsm_sinkhorn_update(n, m, D, x.data_ptr(), y.data_ptr(), dual.data_ptr() if dual is not None else None,
                                        float(log_weight), float(eps), prev.data_ptr() if prev is not None else None, out.data_ptr(),
                                        w.data_ptr() if want_wsum else None, sc.data_ptr(), sc.numel() * 4, C.c_void_p(stream)),
               "epsm_sinkhorn_update")
    return (out, w) if want_wsum else out


def sinkhorn_divergence_and_grad_hip(x: torch.Tensor, y: torch.Tensor, blur: float = 0.01, scaling: float = 0.9):
    """The iteration of ``sinkhorn_divergence`` on the HIP kernel, one call per dual update (h = log-weight + dual / eps and
    the averaging happen inside): returns (divergence, d divergence / d x)."""
    with torch.no_grad():
        x, y = x.detach().float().contiguous(), y.detach().float().contiguous()
        n, m = x.shape[0], y.shape[0]
        la, lb = -math.log(n), -math.log(m)
        mins = torch.minimum(x.min(0).values, y.min(0).values)
        maxs = torch.maximum(x.max(0).values, y.max(0).values)
        diameter = float((maxs - mins).norm().clamp_min(1e-12))
        eps_list = [diameter ** 2]
        e = 2 * math.log(diameter)
        while e > 2 * math.log(blur):
            eps_list.append(math.exp(e)); e += 2 * math.log(scaling)
        eps_list.append(blur ** 2)
        eps = eps_list[0]
        f_aa, g_bb = update_hip(eps, x, x, None, la), update_hip(eps, y, y, None, lb)
        g_ab, f_ba = update_hip(eps, y, x, None, la), update_hip(eps, x, y, None, lb)
        for eps in eps_list:
            f_new = update_hip(eps, x, y, g_ab, lb, prev=f_ba)          # both from the OLD duals (symmetric update)
            g_new = update_hip(eps, y, x, f_ba, la, prev=g_ab)
            f_ba, g_ab = f_new, g_new
            f_aa = update_hip(eps, x, x, f_aa, la, prev=f_aa)
            g_bb = update_hip(eps, y, y, g_bb, lb, prev=g_bb)
        eps = eps_list[-1]
        f_ba_new, w_ba = update_hip(eps, x, y, g_ab, lb, want_wsum=True)
        f_aa_new, w_aa = update_hip(eps, x, x, f_aa, la, want_wsum=True)
        g_ab_new = update_hip(eps, y, x, f_ba, la)                      # from the OLD f_ba (geomloss: one simultaneous assignment)
        g_bb_new = update_hip(eps, y, y, g_bb, lb)
        loss = (f_ba_new - f_aa_new).mean() + (g_ab_new - g_bb_new).mean()
        grad = (w_aa - w_ba) / n                  # d/dx_i of mean_i( softmin over y - softmin over (detached) x )
    return loss, grad


class Matcher:
    """``Matcher(res, device).match_Sinkhorn(render_rgb (res^2,3), gt_rgb (res^2,3)) -> (res^2, 5)``."""

    def __init__(self, res: int, device) -> None:
        self.resolution, self.device = res, torch.device(device)
        lin = torch.linspace(0, 1, res)
        gy, gx = torch.meshgrid(lin, lin, indexing="ij")
        # matcher.py:14-17: pos[..., 0] = x (column), pos[..., 1] = y (row)
        self.pos = torch.stack([gx, gy], dim=2).reshape(-1, 2).to(self.device)
        self.blur, self.scaling = 0.01, 0.9
        self.backend = "hip"                    # on a GPU: csrc/epsm_matcher.hip; "torch": the dense form below (the checker)
        self.num_vectors, self.num_principle_vectors, self.rgb_weight = 50, 3, 1.0       # matcher.py:22-25

    def match_Sinkhorn(self, render_point: torch.Tensor, gt_rgb: torch.Tensor) -> torch.Tensor:
        target = torch.cat([gt_rgb.clamp(0, 1).to(self.device, torch.float32), self.pos], dim=1)      # matcher.py:52-54
        render = torch.cat([render_point.clamp(0, 1).to(self.device, torch.float32), self.pos], dim=1)
        if self.device.type == "cuda" and self.backend == "hip":
            _, g = sinkhorn_divergence_and_grad_hip(render, target, self.blur, self.scaling)
            return g * (self.resolution * self.resolution)                                             # matcher.py:60
        render.requires_grad_(True)
        loss = sinkhorn_divergence(render, target, self.blur, self.scaling)
        (g,) = torch.autograd.grad(loss * self.resolution * self.resolution, [render])                 # matcher.py:60
        return g

    def match_sliced_wasserstein(self, render_point: torch.Tensor, gt_rgb: torch.Tensor, generator=None,
                                 pca_V: torch.Tensor = None, directions: torch.Tensor = None) -> torch.Tensor:
        """matcher.py:76-116 (the reference's geomloss-free alternative): both point clouds (r,g,b,x,y) are
        projected onto ``num_vectors`` random unit directions of the (principal colour axes of the target + position)
        space, each projection is matched by sorting, and the gradient of the summed squared differences w.r.t. the
        rendered points is returned, (res^2, 5).

        The gradient is written in closed form (the reference differentiates the sorted projections with autograd):
        with P = X D the projections of the rendered points X onto the directions D and e the differences of the
        sorted columns, dL/dP puts 2 e back at the rows the sort took them from and dL/dX = (dL/dP) D^T, the colour
        part going back through the PCA basis.  Random draws happen in the reference's order (``torch.pca_lowrank``,
        then ``torch.rand`` for the directions), so a call under the same torch seed reproduces the reference --
        pinned by tests/golden/matcher_sliced_*.npz (tests/test_matcher.py); ``pca_V`` / ``directions`` inject the
        draws instead."""
        w = self.rgb_weight
        target = torch.cat([gt_rgb.clamp(0, 1).to(self.device, torch.float32) * w, self.pos], dim=1)
        render = torch.cat([render_point.clamp(0, 1).to(self.device, torch.float32).detach() * w, self.pos], dim=1)
        q = self.num_principle_vectors
        if q > 0:
            assert q <= 3
            if pca_V is None:
                _, _, pca_V = torch.pca_lowrank(target[:, :3], q=3)                  # colour axes of the TARGET
            Vp = pca_V.to(self.device, torch.float32)[:, :q]
            t_pts = torch.cat([target[:, :3] @ Vp, target[:, 3:]], dim=1)
            r_pts = torch.cat([render[:, :3] @ Vp, render[:, 3:]], dim=1)
        else:
            Vp, t_pts, r_pts = None, target, render
        dim = 2 + (q if q > 0 else 3)
        if directions is None:
            directions = torch.rand((dim, self.num_vectors), device=self.device, generator=generator)
        dirs = torch.nn.functional.normalize(directions.to(self.device, torch.float32) * 2.0 - 1.0, p=2, dim=0)
        pr, order = torch.sort(r_pts @ dirs, dim=0, stable=True)
        pt, _ = torch.sort(t_pts @ dirs, dim=0, stable=True)
        gP = torch.zeros_like(pr).scatter_(0, order, 2.0 * (pr - pt))            # d/dP of sum (sorted_r - sorted_t)^2
        gX = gP @ dirs.t()                                                        # (N, dim)
        g_rgb = gX[:, :q] @ Vp.t() if q > 0 else gX[:, :3]
        g = torch.cat([g_rgb, gX[:, -2:]], dim=1)
        g[:, :3] /= w
        return g
