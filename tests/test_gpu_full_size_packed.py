"""The EXACT kernel instantiation bench.py times -- ``epsm_backward_cp_kernel<variant, tangents in kernel, packed log, fixed-
point rows, windows of 2048>`` behind ``epsm_backward_pass_packed`` -- on ONE slab built exactly as bench.py builds it
(1024 x 1024 film @ 256 spp, 2^24 paths = 64 image rows, K = 5, V = 100 000, ``PackedLog``), for the headline configuration
(bathroom, ``manifold``: BASELINE.json configs[3]) and for the pool-caustic slab of configs[2] (``manifold_caustic``).
Reference shape: epsm.py:84-306 at N = res^2 spp.

The float64 oracle needs minutes per million paths, so at 2^24 paths the checks are the size-independent ones --
run-to-run agreement and the reference-shaped three stages (tangent kernel -> calc_grad lists -> scatter kernel) as an
independent route to the same sums -- plus a SPOT CHECK against the oracle on 4 096 paths cut out of the slab (16 pixels),
run through the same instantiation (windows of 2048) with private parameter rows per (path, vertex), tests/_per_path.py."""
import pytest
import torch

from _per_path import check_private_rows, private_addressing
from _util import assert_two_routes_agree

pytestmark = pytest.mark.gpu

SLAB_PATHS = 1 << 24


def _cut(trace, a, n):
    """Paths [a, a + n) of a PathTrace as a PathTrace of its own (per-field records; ``path_offset`` keeps the pixel mapping)."""
    import epsm_mitsuba3_amd as epsm
    cut = lambda t: None if t is None else t[a:a + n].contiguous()
    pi = [{"cam": cut(trace.path_info[0]["cam"])}]
    for r in trace.path_info[1:]:
        pi.append({k: ([cut(x) for x in v] if isinstance(v, (list, tuple)) else (cut(v) if torch.is_tensor(v) else v)) for k, v in r.items()})
    return epsm.PathTrace(res=trace.res, spp=trace.spp, ray_o=cut(trace.ray_o), ray_d=cut(trace.ray_d), ray_dx=cut(trace.ray_dx),
                          ray_dy=cut(trace.ray_dy), path_info=pi, scatter_info=None, path_offset=trace.path_offset + a,
                          n_paths_total=trace.n_paths_total)


# ("manifold", "specular"): bench.py's dense_specular slab -- no diffuse vertex, ~40 rows per path, most of which find no slot in the
# LDS table and leave four lanes per row (epsm_wave_scatter.h, drain_queue): the slab that exercises that way out at full size
@pytest.mark.parametrize("variant,profile", [("manifold", "bathroom"), ("manifold_caustic", "pool"), ("manifold", "specular")])
def test_headline_slab_on_the_timed_kernel(variant, profile):
    import epsm_mitsuba3_amd as epsm
    from epsm_mitsuba3_amd import _lib
    from epsm_mitsuba3_amd.records import PackedLog, PackedRecords, PackedScatter, num_param_grads
    dev = torch.device("cuda", 0)
    res, spp, K, V, B = 1024, 256, 5, 100000, 4                  # bench.py CONFIGS[0] / CONFIGS[3]
    N = SLAB_PATHS
    assert _lib.lib().epsm_get_option(_lib.OPT_SMALL_WAVEFRONT_PATHS) < N      # the large form: windows of 2048 paths
    scene = epsm.SyntheticScene(res=res, n_vertices=K, n_scene_vertices=V, n_bsdfs=B, profile=profile, device=dev, tile_paths=N)
    trace = scene.tile(0, 0, N, seed=0, spp=spp, K=K, lean=True)               # bench.py: slab 0
    log = PackedLog.from_trace(trace, device=dev, table=scene.triangle_table(), free=False)
    g = torch.Generator(device=dev).manual_seed(1)
    grad_in = torch.randn((res, res, 5), generator=g, device=dev) * 1e-3       # bench.py's gradient image
    integ = epsm.load_dict({"type": variant, "max_depth": 8})

    # ---- (1) the timed route twice: the order of the float atomics only
    runs = []
    for _ in range(2):
        params = epsm.ParamGrads(V, B, device=dev)
        integ.backward_from_trace(trace, params, grad_in, packed=log)
        torch.cuda.synchronize()
        assert bool(torch.isfinite(params.flat).all())
        runs.append(params.flat.double().cpu())
    m = float(runs[0].abs().max())
    assert m > 0
    assert float((runs[0] - runs[1]).abs().max()) <= 1e-5 * m

    # ---- (1b) the same launch WITHOUT the camera-origin sum (grad_o_sum = NULL: what a log traced under EPSM_TRACE_FUSE_FIRST_HIT gets):
    # paths without a term take no lane (the kernel's DROP instantiation) at full size.  Same rows.
    from epsm_mitsuba3_amd.tangent_scatter import backward_pass_packed
    nocam = epsm.ParamGrads(V, B, device=dev)
    backward_pass_packed(variant, log, grad_in, spp, res, nocam.pos, nocam.nrm, nocam.alpha, None, clip=0.1, path_offset=trace.path_offset)
    torch.cuda.synchronize()
    withcam = epsm.ParamGrads(V, B, device=dev)
    withcam.flat.copy_(runs[0].float())
    for a, b, name in ((nocam.pos, withcam.pos, "pos"), (nocam.nrm, withcam.nrm, "nrm"), (nocam.alpha, withcam.alpha, "alpha")):
        assert float((a.double() - b.double()).abs().max()) <= 1e-5 * m, name
    assert float(nocam.cam_origin.abs().max()) == 0
    del nocam, withcam

    # ---- (2) the reference's three stages on the same records: dense calc_grad lists + scatter kernel (another restatement
    # of the per-path arithmetic: csrc/epsm_path_core.h), with the clamp moved by -+2 % for the threshold allowance
    packed = (PackedRecords(trace.path_info, device=dev), PackedScatter(trace.scatter_info, device=dev, table=scene.triangle_table()))
    P = num_param_grads(variant, K)
    out = (torch.empty((P, N, 3), device=dev), torch.empty((K, N, 3), device=dev), torch.empty((K, N, 3), device=dev))
    stages = []
    for clip in (0.1, 0.098, 0.102):
        three = epsm.load_dict({"type": variant, "max_depth": 8, "fused": False, "outlier_clip": clip})
        params = epsm.ParamGrads(V, B, device=dev)
        three.backward_from_trace(trace, params, grad_in, packed=packed, out=out)
        torch.cuda.synchronize()
        stages.append(params.flat.double().cpu())
    rep = assert_two_routes_agree(runs[0], *stages, name=f"{variant} {profile} slab")
    print(variant, profile, "one launch on the packed log vs three stages:", rep)
    assert abs(float(runs[0].sum() - stages[0].sum())) <= 1e-4 * float(runs[0].abs().sum()) + float((stages[1] - stages[2]).abs().sum())
    del out, packed, stages

    # ---- (3) oracle spot check: 4 096 paths of the slab (16 pixels of row 37), private rows, same instantiation
    n, a = 4096, (37 * res + 500) * spp
    sub = _cut(trace, a, n)
    gen = torch.Generator().manual_seed(3)
    table, si = private_addressing(n, K, dev, gen)
    sub.scatter_info = si
    small_grad = grad_in * 2e-2                                   # small tangents: few components near the +-0.1 clamp
    params = epsm.ParamGrads(6 * n * K, n * K, device=dev)
    with _lib.options(small_wavefront_paths=0):                  # windows of 2048 paths for 4 096 paths as well
        integ.backward_from_trace(sub, params, small_grad, packed=PackedLog.from_trace(sub))
    torch.cuda.synchronize()
    check_private_rows(variant, sub, si, params, small_grad, K, label=f"{profile} slab, paths {a}..{a + n}")
