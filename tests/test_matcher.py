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
