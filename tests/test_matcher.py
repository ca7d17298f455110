"""Property tests of the plain-torch Sinkhorn matcher (EPSM/utils/matcher.py:51-63 restated without
geomloss; parity unpinned -- see epsm_mitsuba3_amd/matcher.py)."""
import torch

from epsm_mitsuba3_amd.matcher import Matcher, sinkhorn_divergence


def test_identical_clouds_have_zero_divergence_and_gradient():
    g = torch.Generator().manual_seed(0)
    x = torch.rand((200, 5), generator=g).requires_grad_(True)
    s = sinkhorn_divergence(x, x.detach().clone())
    (gr,) = torch.autograd.grad(s, [x])
    assert abs(float(s)) < 1e-6 and float(gr.abs().max()) < 1e-5


def test_translated_cloud_gradient_is_the_displacement():
    """For blur -> 0 the divergence is W2^2/2 and d/dx_i = (x_i - T(x_i))/N; times N: the displacement."""
    g = torch.Generator().manual_seed(1)
    y = torch.rand((300, 5), generator=g)
    shift = torch.tensor([0.0, 0.0, 0.0, 0.06, -0.04])
    x = (y + shift).requires_grad_(True)
    s = sinkhorn_divergence(x, y)
    (gr,) = torch.autograd.grad(s * x.shape[0], [x])
    assert abs(float(s) - 0.5 * float(shift.pow(2).sum())) < 2e-4
    err = (gr - shift).norm(dim=1)
    assert float(err.median()) < 5e-3 and float(err.mean()) < 1e-2


def test_descent_on_the_matcher_gradient_moves_a_blob_onto_the_target():
    res = 16
    m = Matcher(res, "cpu")
    yy, xx = torch.meshgrid(torch.linspace(0, 1, res), torch.linspace(0, 1, res), indexing="ij")
    blob = lambda cx, cy: torch.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / 0.01)[..., None].repeat(1, 1, 3).reshape(-1, 3)
    gt = blob(0.7, 0.5)
    grad = m.match_Sinkhorn(blob(0.3, 0.5), gt)
    assert grad.shape == (res * res, 5)
    # bright pixels of the rendered blob are asked to move in +x (negative gradient = descent direction)
    bright = blob(0.3, 0.5)[:, 0] > 0.5
    assert float(grad[bright, 3].mean()) < -0.05 and abs(float(grad[bright, 4].mean())) < 0.05


def test_sliced_wasserstein_moves_a_blob_towards_its_target():
    """match_sliced_wasserstein (matcher.py:76-116): zero for identical images; for a shifted blob the position
    gradient of the bright pixels points from where the blob is to where it should be (descent direction = -g)."""
    from epsm_mitsuba3_amd.matcher import Matcher
    res = 24
    m = Matcher(res, "cpu")
    yy, xx = torch.meshgrid(torch.arange(res), torch.arange(res), indexing="ij")
    blob = lambda cx, cy: torch.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / 8.0)[..., None].repeat(1, 1, 3).reshape(-1, 3).float()
    a, b = blob(7, 12), blob(15, 12)                                   # target is 8 pixels to the right
    gen = torch.Generator().manual_seed(0)
    g_same = m.match_sliced_wasserstein(a, a, generator=gen)
    assert g_same.shape == (res * res, 5) and float(g_same.abs().max()) < 1e-6
    g = m.match_sliced_wasserstein(a, b, generator=torch.Generator().manual_seed(0))
    bright = a[:, 0] > 0.5
    assert float(g[bright, 3].mean()) < 0                               # -g points to +x
    assert abs(float(g[bright, 4].mean())) < 0.3 * abs(float(g[bright, 3].mean()))


import glob
import os

import numpy as np
import pytest

_SLICED = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "matcher_sliced_*.npz")))


@pytest.mark.parametrize("path", _SLICED, ids=[os.path.basename(p) for p in _SLICED])
def test_sliced_wasserstein_matches_the_reference(path):
    """``match_sliced_wasserstein`` against the reference's own function (EPSM/utils/matcher.py:76-116) run in place
    by tests/golden/gen_matcher_golden.py: (1) with the reference's random draws injected -- the closed-form gradient
    against the reference's autograd; (2) under the same torch seed -- the draws are made in the same order."""
    z = np.load(path)
    res = int(z["res"])
    m = Matcher(res, "cpu")
    assert m.num_vectors == int(z["num_vectors"]) and m.num_principle_vectors == int(z["num_principle_vectors"])
    rd, gt, want = torch.from_numpy(z["render"]), torch.from_numpy(z["target"]), torch.from_numpy(z["grad"])
    g = m.match_sliced_wasserstein(rd, gt, pca_V=torch.from_numpy(z["pca_V"]), directions=torch.from_numpy(z["rand"]))
    scale = float(want.abs().max())
    assert tuple(g.shape) == (res * res, 5) and scale > 0.1
    assert float((g - want).abs().max()) <= 2e-5 * scale
    torch.manual_seed(int(z["seed"]))
    g2 = m.match_sliced_wasserstein(rd, gt)
    assert float((g2 - want).abs().max()) <= 1e-3 * scale


def test_sliced_wasserstein_gradient_is_the_autograd_gradient():
    g0 = torch.Generator().manual_seed(5)
    res = 12
    m = Matcher(res, "cpu")
    rd, gt = torch.rand((res * res, 3), generator=g0), torch.rand((res * res, 3), generator=g0)
    V = torch.linalg.qr(torch.randn((3, 3), generator=g0))[0]
    D = torch.rand((5, m.num_vectors), generator=g0)
    g = m.match_sliced_wasserstein(rd, gt, pca_V=V, directions=D)
    x = torch.cat([rd.clamp(0, 1), m.pos], dim=1).requires_grad_(True)
    t = torch.cat([gt.clamp(0, 1), m.pos], dim=1)
    dirs = torch.nn.functional.normalize(D * 2 - 1, p=2, dim=0)
    proj = lambda p: torch.cat([p[:, :3] @ V, p[:, 3:]], dim=1) @ dirs
    loss = ((torch.sort(proj(x), dim=0, stable=True)[0] - torch.sort(proj(t), dim=0, stable=True)[0]) ** 2).sum()
    (want,) = torch.autograd.grad(loss, [x])
    assert torch.allclose(g, want, rtol=1e-4, atol=1e-6)
