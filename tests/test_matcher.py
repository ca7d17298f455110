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
