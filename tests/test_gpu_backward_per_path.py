"""First-hand, PER-PATH parity of the backward kernel the benchmark times (``epsm_backward_pass_packed``: first-vertex
tangent + calc_grad + scatter in one launch on the native log, csrc/epsm_backward_cp.hip) against the float64 oracle.

The kernel never writes calc_grad's per-path lists -- it adds rows into the parameter buffers -- so the test gives every
(path, vertex) PRIVATE rows (tests/_per_path.py: its own hit triangle, its own emitter triangle, its own alpha slot); the
buffers then hold, row by row, the per-path quantities the reference's replay would scatter, and are compared path by
path with the oracle's lists under the conditioning gate of SURVEY.md 8c -- the same yardstick as for the dense calc_grad
kernel (test_gpu_parity.py), with the tangent of epsm.py:250-272 computed in the kernel."""
import pytest
import torch

from _per_path import check_private_rows, private_addressing

pytestmark = pytest.mark.gpu


@pytest.mark.usefixtures("window_form")
@pytest.mark.parametrize("kind,profile", [("manifold", "bathroom"), ("manifold", "specular"), ("manifold", "mixed"),
                                          ("manifold_caustic", "pool"), ("manifold_caustic", "mixed")])
@pytest.mark.parametrize("K", [1, 2, 3, 5])
@pytest.mark.parametrize("layout", ["packed log", "per-field records"])
def test_backward_pass_per_path(kind, profile, K, layout):
    import epsm_mitsuba3_amd as epsm
    from epsm_mitsuba3_amd.records import PackedLog
    dev = torch.device("cuda", 0)
    res, spp = 50, 8
    N = res * res * spp                      # 20 000 paths
    scene = epsm.SyntheticScene(res=res, n_vertices=K, n_scene_vertices=3000, n_bsdfs=4, profile=profile, device=dev, tile_paths=N)
    trace = scene.tile(0, 0, N, seed=17 + K, spp=spp, K=K)
    gen = torch.Generator().manual_seed(3)
    table, si = private_addressing(N, K, dev, gen)
    trace.scatter_info = si
    grad_in = (torch.randn((res, res, 5), generator=gen) * 2e-5).to(dev)      # small tangents: few components near the +-0.1 clamp
    V, B = 6 * N * K, N * K
    params = epsm.ParamGrads(V, B, device=dev)
    integ = epsm.load_dict({"type": kind, "max_depth": 8})
    integ.backward_from_trace(trace, params, grad_in, packed=PackedLog.from_trace(trace) if layout == "packed log" else None)
    torch.cuda.synchronize()
    check_private_rows(kind, trace, si, params, grad_in, K, label=f"{profile} {layout}")


@pytest.mark.usefixtures("window_form")
def test_unbounded_emitter_weights_leave_finite_rows():
    """the accumulating kernel sums its LDS rows in 64-bit fixed point (csrc/epsm_wave_scatter.h, AccFixed64), and
    the clamp of calc_grad bounds a term only BEFORE the replay multiplies it by the emitter weight (epsm.py:622-627) -- an
    unbounded factor.  A non-finite product must add nothing (the reference's buffer would hold NaN there, which its loop
    scrubs to 0, EPSM/optim.py:143-154), a finite one beyond the rows' range must saturate with its own sign -- never wrap
    around or turn into a small finite number -- and no other row may notice."""
    import epsm_mitsuba3_amd as epsm
    from epsm_mitsuba3_amd.records import PackedLog
    dev = torch.device("cuda", 0)
    res, spp, K = 32, 8, 3
    N = res * res * spp
    scene = epsm.SyntheticScene(res=res, n_vertices=K, n_scene_vertices=3000, n_bsdfs=4, profile="specular", device=dev, tile_paths=N)
    trace = scene.tile(0, 0, N, seed=5, spp=spp, K=K)
    gen = torch.Generator().manual_seed(9)
    table, si = private_addressing(N, K, dev, gen)
    grad_in = (torch.randn((res, res, 5), generator=gen) * 1e-3).to(dev)
    integ = epsm.load_dict({"type": "manifold", "max_depth": 8})

    def run(weights):
        for k in range(K):
            si[k]["emit"][:, 3] = weights[k].to(dev).contiguous().view(torch.int32)
        trace.scatter_info = si
        params = epsm.ParamGrads(6 * N * K, N * K, device=dev)
        integ.backward_from_trace(trace, params, grad_in, packed=PackedLog.from_trace(trace))
        torch.cuda.synchronize()
        return params

    clean_w = [si[k]["_f"][2].float() for k in range(K)]
    clean = run(clean_w)
    idx = torch.arange(N)
    inf_paths, big_paths = (idx % 7) == 0, (idx % 7) == 3
    bad_w = [torch.where(inf_paths, torch.full_like(w, float("inf")), torch.where(big_paths, torch.full_like(w, 1e30), w)) for w in clean_w]
    bad = run(bad_w)
    assert bool(torch.isfinite(bad.flat).all())
    c = clean.pos.view(2, K, N, 3, 3).cpu()
    b = bad.pos.view(2, K, N, 3, 3).cpu()
    untouched = ~(inf_paths | big_paths)
    assert float(c[1].abs().max()) > 0                                       # there ARE light-sample terms
    same = lambda x, y: torch.allclose(x, y, rtol=1e-5, atol=1e-7 * float(y.abs().max()))      # (order of the float atomics of a crowded table)
    assert same(b[0], c[0]) and same(bad.nrm, clean.nrm) and same(bad.alpha, clean.alpha)   # hit rows, normals, alphas
    assert same(b[1][:, untouched], c[1][:, untouched])                      # emitter rows of the other paths
    assert float(b[1][:, inf_paths].abs().max()) == 0.0                      # inf x term: nothing (0 x inf = NaN included)
    live = c[1][:, big_paths] != 0
    got, want = b[1][:, big_paths][live], c[1][:, big_paths][live]
    # saturated at the top of the rows' range (or, for a row the crowded table sent straight to HBM, the product itself),
    # with the term's own sign: never wrapped around, never a small number
    assert live.any() and bool((got.sign() == want.sign()).all())
    assert float(got.abs().min()) > 5e5
    assert bool((b[1][:, big_paths][~live] == 0).all())
