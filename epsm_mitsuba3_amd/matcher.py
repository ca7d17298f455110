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
    # last extrapolation with the graph attached to x (duals detached)
    f_ba = _softmin(eps, _cost(x, yd), b_log + g_ab / eps)
    f_aa_new = _softmin(eps, _cost(x, xd), a_log + f_aa / eps)
    with torch.no_grad():
        g_ab_new = _softmin(eps, C_yx, a_log + f_ba.detach() / eps)
        g_bb_new = _softmin(eps, C_yy, b_log + g_bb / eps)
    return (f_ba - f_aa_new).mean() + (g_ab_new - g_bb_new).mean()


class Matcher:
    """``Matcher(res, device).match_Sinkhorn(render_rgb (res^2,3), gt_rgb (res^2,3)) -> (res^2, 5)``."""

    def __init__(self, res: int, device) -> None:
        self.resolution, self.device = res, torch.device(device)
        lin = torch.linspace(0, 1, res)
        gy, gx = torch.meshgrid(lin, lin, indexing="ij")
        # matcher.py:14-17: pos[..., 0] = x (column), pos[..., 1] = y (row)
        self.pos = torch.stack([gx, gy], dim=2).reshape(-1, 2).to(self.device)
        self.blur, self.scaling = 0.01, 0.9
        self.num_vectors, self.num_principle_vectors, self.rgb_weight = 50, 3, 1.0       # matcher.py:22-25

    def match_Sinkhorn(self, render_point: torch.Tensor, gt_rgb: torch.Tensor) -> torch.Tensor:
        target = torch.cat([gt_rgb.clamp(0, 1).to(self.device, torch.float32), self.pos], dim=1)      # matcher.py:52-54
        render = torch.cat([render_point.clamp(0, 1).to(self.device, torch.float32), self.pos], dim=1).requires_grad_(True)
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
