"""epsm_sinkhorn_softmin (csrc/epsm_matcher.hip) against the dense torch form of epsm_mitsuba3_amd/matcher.py -- the
"plain PyTorch fp32 reference of the same op" -- and the matcher built on it against the matcher built on the dense form."""
import math
import time

import pytest
import torch

pytestmark = pytest.mark.gpu


def _clouds(n, m, D, seed, dev):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand((n, D), generator=g)
    y = torch.rand((m, D), generator=g) * 0.9 + 0.05
    h = torch.randn((m,), generator=g) * 3.0 - math.log(m)
    return x.to(dev), y.to(dev), h.to(dev)


@pytest.mark.parametrize("n,m,D", [(4096, 4096, 5), (1000, 3333, 5), (257, 1, 3), (1, 700, 7), (5000, 256, 1), (0, 10, 5)])
def test_softmin_matches_the_dense_form(n, m, D):
    from epsm_mitsuba3_amd.matcher import softmin_hip
    dev = torch.device("cuda", 0)
    x, y, h = _clouds(n, m, D, n + m, dev)
    for eps in (1.7 ** 2, 0.1, 1e-2, 1e-4):                       # diameter^2 ... blur^2 of the matcher's schedule
        out, w = softmin_hip(eps, x, y, h, want_wsum=True)
        out2 = softmin_hip(eps, x, y, h)
        assert out.shape == (n,) and w.shape == (n, D)
        if n == 0:
            continue
        C = 0.5 * torch.cdist(x.double(), y.double()).pow(2)
        z = h.double()[None, :] - C / eps
        ref = -eps * torch.logsumexp(z, dim=1)
        wref = torch.softmax(z, dim=1) @ y.double()
        scale = float(ref.abs().max()) + eps
        assert float((out.double() - ref).abs().max()) <= 2e-5 * scale + 2e-6, (n, m, D, eps)
        assert torch.equal(out, out2)                             # the weighted sums do not change the value
        # where the softmax is sharp (small eps) a tie between two columns is decided in fp32: compare through the cost
        assert float((w.double() - wref).abs().max()) <= 1e-3, (n, m, D, eps)


def test_softmin_rejects_bad_arguments():
    from epsm_mitsuba3_amd import _lib
    from epsm_mitsuba3_amd.matcher import softmin_hip
    dev = torch.device("cuda", 0)
    x, y, h = _clouds(16, 16, 5, 0, dev)
    with pytest.raises(_lib.EpsmError):
        softmin_hip(0.0, x, y, h)
    lib = _lib.lib()
    assert lib.epsm_sinkhorn_softmin(16, 16, 9, x.data_ptr(), y.data_ptr(), h.data_ptr(), 1.0, h.data_ptr(), None, h.data_ptr(), 4, None) != 0
    assert lib.epsm_sinkhorn_softmin(16, 16, 5, x.data_ptr(), y.data_ptr(), h.data_ptr(), 1.0, h.data_ptr(), None, h.data_ptr(), 4, None) != 0   # scratch too small
    assert lib.epsm_sinkhorn_splits(65536, 65536) >= 4 and lib.epsm_sinkhorn_scratch_bytes(256, 256, 5) == 1 * 256 * 7 * 4


@pytest.mark.parametrize("res", [16, 48, 128])          # 128: the matching resolution of most of the reference's experiments
def test_matcher_on_the_kernel_equals_the_dense_matcher(res):
    from epsm_mitsuba3_amd.matcher import Matcher
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(res)
    render = torch.rand((res * res, 3), generator=g).to(dev)
    gt = torch.rand((res * res, 3), generator=g).to(dev)
    m = Matcher(res, dev)
    assert m.backend == "hip"
    a = m.match_Sinkhorn(render, gt)
    m.backend = "torch"
    b = m.match_Sinkhorn(render, gt)
    scale = float(b.abs().max())
    assert scale > 0 and float((a - b).abs().max()) <= 2e-3 * scale, float((a - b).abs().max()) / scale


def test_matcher_at_the_reference_resolutions():
    """match_res = 128 (most of the reference's experiments) and 256 (human, glassslab): 16 384 and 65 536 points.  The
    dense form needs four 1 GB / 17 GB matrices; the kernel needs the clouds.  Prints the time per call."""
    from epsm_mitsuba3_amd.matcher import Matcher
    dev = torch.device("cuda", 0)
    for res in (128, 256):
        g = torch.Generator().manual_seed(res)
        render = torch.rand((res * res, 3), generator=g).to(dev)
        gt = (render + 0.05 * torch.randn((res * res, 3), generator=g).to(dev)).clamp(0, 1)
        m = Matcher(res, dev)
        m.match_Sinkhorn(render, gt)
        torch.cuda.synchronize(); t = time.perf_counter()
        grad = m.match_Sinkhorn(render, gt)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        print(f"match_Sinkhorn at {res} x {res}: {dt * 1e3:.1f} ms")
        assert grad.shape == (res * res, 5) and bool(torch.isfinite(grad).all()) and float(grad.abs().max()) > 0
        assert dt < (3.0 if res == 256 else 1.0)


def test_update_is_the_softmin_with_h_and_averaging_inside():
    from epsm_mitsuba3_amd.matcher import softmin_hip, update_hip
    dev = torch.device("cuda", 0)
    x, y, dual = _clouds(3000, 2000, 5, 7, dev)
    prev = torch.randn(3000, device=dev)
    for eps in (1.0, 1e-3):
        lw = -math.log(2000)
        ref = softmin_hip(eps, x, y, lw + dual / eps)
        assert torch.allclose(update_hip(eps, x, y, dual, lw), ref, rtol=1e-5, atol=1e-6)
        assert torch.allclose(update_hip(eps, x, y, dual, lw, prev=prev), 0.5 * (prev + ref), rtol=1e-5, atol=1e-6)
        assert torch.allclose(update_hip(eps, x, y, None, lw), softmin_hip(eps, x, y, torch.full((2000,), lw, device=dev)), rtol=1e-5, atol=1e-6)
        o, w = update_hip(eps, x, y, dual, lw, want_wsum=True)
        o2, w2 = softmin_hip(eps, x, y, lw + dual / eps, want_wsum=True)
        assert torch.allclose(w, w2, rtol=1e-4, atol=1e-5)
