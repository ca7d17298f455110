"""EPSM_TRACE_GRADIENT_ONLY (include/epsm_trace.h): the backward trace retires a path once nothing behind its last logged
vertex can reach calc_grad.  (1) the rule against the term masks themselves (cp::manifold_plan / caustic_plan =
epsm.py:793-803, 852-856, 916-921, 998-999, 1172-1183) over every flag word of up to three vertices and random longer ones;
(2) the host build of the tracer with and without the flag: identical logs up to the retirement point, and the float64
oracle pipeline gives the same parameter gradients from both."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests import _scenes as TS
from tests.host_core import lib as host_core


def _plan_fns():
    l = host_core("cp")
    l.epsm_host_plan.restype = C.c_uint32
    l.epsm_host_plan.argtypes = [C.c_int, C.c_uint32]
    l.epsm_host_gradient_live.restype = C.c_int
    l.epsm_host_gradient_live.argtypes = [C.c_uint32, C.c_int, C.c_int]
    return l.epsm_host_plan, l.epsm_host_gradient_live


def _retired_word(w, K, caustic, live):
    """The flag word the gradient-only trace leaves: vertices behind the first k with !live(w, k) are never logged."""
    for k in range(1, K + 1):
        if not ((w >> (5 * (k - 1))) & 4):           # the path itself ended before vertex k: nothing more is logged anyway
            return w & ((1 << (5 * (k - 1))) - 1) | (w & (0x1F << (5 * (k - 1)))), k
        if not live(w & ((1 << (5 * k)) - 1), k, caustic):
            return w & ((1 << (5 * k)) - 1), k
    return w, K


@pytest.mark.parametrize("variant", [0, 1])
def test_retirement_rule_never_changes_the_plan(variant):
    plan, live = _plan_fns()
    rng = np.random.default_rng(5)
    words = [w for w in range(1 << 15)]                                   # every word of up to three vertices
    words += [int(x) for x in rng.integers(0, 1 << 25, size=200000)]     # and random five-vertex ones
    # bias towards long live chains: mesh + active set, few diffuse bits
    base = 0
    for k in range(5):
        base |= (4 | 16) << (5 * k)
    words += [int(base | (x & 0x0B5AD6B)) for x in rng.integers(0, 1 << 25, size=200000)]   # keeps bits 0,1,3 of each vertex random
    n_retired = 0
    for w in words:
        # a path's `active` bits are monotone (epsm.py:735); words that violate this never occur
        act = [(w >> (5 * k + 2)) & 1 for k in range(5)]
        if any(act[k + 1] and not act[k] for k in range(4)):
            continue
        wr, k = _retired_word(w, 5, variant, live)
        n_retired += wr != w
        assert plan(variant, w) == plan(variant, wr), (hex(w), hex(wr), k)
    assert n_retired > 1000


def _scene(tracer):
    from epsm_mitsuba3_amd.exp import clutter
    scene = TS.on_host(clutter.load_scene(device="cpu", n_spheres=8, res=24, spp=4))
    for name in ["floor", "light"] + [f"s{i}" for i in range(8)]:
        scene.attach(name, positions=True, normals=name.startswith("s"))
    scene.tracer = tracer
    return scene


def _trace(scene, variant, gradient_only, packed):
    kw = dict(sensor=2, seed=3, spp=4, max_depth=6, max_log_depth=5, sparse_log=packed, packed_log=packed)
    if gradient_only:
        kw["gradient_only"] = variant
    return scene.trace_paths(**kw)


@pytest.mark.parametrize("tracer", ["mega", "wavefront"])
@pytest.mark.parametrize("variant", ["manifold", "manifold_caustic"])
def test_host_tracer_logs_agree_up_to_the_retirement_point(variant, tracer):
    scene = _scene(tracer)
    plan, live = _plan_fns()
    (full,) = _trace(scene, variant, False, True)
    (cut,) = _trace(scene, variant, True, True)
    wf, wc = full.log.flags.numpy().astype(np.uint32), cut.log.flags.numpy().astype(np.uint32)
    caustic = int(variant == "manifold_caustic")
    n_short = 0
    for i in range(wf.shape[0]):
        wr, k = _retired_word(int(wf[i]), full.log.K, caustic, live)
        # (a vertex that retires its path by the rule is only looked at: no emitter sample is drawn for it, its active_em bit
        # -- which no term reads there -- stays 0, and only the first sector of its record is written)
        by_rule = not live(wr & ((1 << (5 * k)) - 1), k, caustic)           # (a missed ray is such a vertex too)
        em_bit = (8 << (5 * (k - 1))) if by_rule else 0
        assert int(wc[i]) | em_bit == wr | em_bit and not (int(wc[i]) & em_bit), (i, hex(int(wf[i])), hex(int(wc[i])), hex(wr))
        n_short += wr != int(wf[i])
        # the records that exist in the cut log are bit for bit those of the full trace -- but for word 27, the emitter weight
        # eweight = sum Lr_dir, where no term reads it (its visibility ray is not traced then): a light-sampling term wN(k),
        # bit k - 1 of the manifold plan; manifold_caustic has none
        nk = max(1, sum(1 for j in range(full.log.K) if (wr >> (5 * j)) & 0x1F))
        a, b = full.log.verts[i, :nk].view(torch.int32).clone(), cut.log.verts[i, :nk].view(torch.int32).clone()
        p = plan(caustic, wr)
        for j in range(nk):
            if caustic or not ((p >> j) & 1):
                a[j, 27] = b[j, 27] = 0
        if by_rule and k <= nk:                                    # the retiring vertex: first sector only, eta (word 15) not drawn
            a[k - 1, 15:] = 0; b[k - 1, 15:] = 0
        assert torch.equal(a, b), i
        assert plan(caustic, int(wc[i])) == plan(caustic, int(wf[i])), i          # the words differ, the terms they stand for do not
    assert n_short >= 40, "the scene retires too few paths to test anything"     # (an open scene: most paths leave it by themselves)
    assert torch.equal(full.log.rays, cut.log.rays)


@pytest.mark.parametrize("variant", ["manifold", "manifold_caustic"])
def test_oracle_pipeline_gives_the_same_gradients_from_both_traces(variant):
    from tests._pipeline_oracle import oracle_backward
    scene = _scene("wavefront")
    g = torch.Generator().manual_seed(2)
    grad_in = torch.randn((24, 24, 5), generator=g) * 1e-3
    full = _trace(scene, variant, False, False)
    cut = _trace(scene, variant, True, False)
    a = oracle_backward(variant, full, grad_in, scene.V, len(scene.alpha_slots))
    b = oracle_backward(variant, cut, grad_in, scene.V, len(scene.alpha_slots))
    assert float(a[0].abs().max()) > 0
    for x, y in zip(a, b):
        assert torch.allclose(x, y, rtol=1e-12, atol=1e-18)        # the same float64 terms (up to the order of the OpenMP sums)
    # and the cut trace did stop early: fewer active vertices in its log
    act = lambda trs: sum(int(r["active"].sum()) for tr in trs for r in tr.path_info[1:])
    assert act(cut) < act(full) - 40


@pytest.mark.parametrize("variant", ["manifold", "manifold_caustic"])
def test_first_hit_fusion_on_the_host_build(variant):
    """EPSM_TRACE_FUSE_FIRST_HIT (include/epsm_trace.h) on the host build of the wavefront tracer: a path the rule retires at its first
    vertex leaves flag word 0 and nothing else in the log; every other path's rays, flag word and records are those of the unfused
    trace; and what the stage added to the buffers is what the float64 oracle tangent gives -- d / d ray.o = -sum grad_d over ALL
    paths (epsm.py:255-261), clamp(dldp) b_j on the vertex rows of a diffuse first hit (epsm.py:561-562, 791-792, 932-944)."""
    from oracle.binding import oracle_first_vertex_tangent
    scene = _scene("wavefront")
    _, live = _plan_fns()
    caustic = int(variant == "manifold_caustic")
    res, spp = 24, 4
    g = torch.Generator().manual_seed(4)
    grad_in = (torch.randn((res, res, 5), generator=g) * 1e-2).contiguous()
    params = scene.param_grads()
    kw = dict(sensor=2, seed=3, spp=spp, max_depth=6, max_log_depth=5, sparse_log=True, packed_log=True, gradient_only=variant)
    (cut,) = scene.trace_paths(**kw)
    (fus,) = scene.trace_paths(**kw, first_hit=(grad_in, params, 0.1, True))
    assert fus.log.first_hit_done and not cut.log.first_hit_done
    wc, wf = cut.log.flags.numpy().astype(np.uint32), fus.log.flags.numpy().astype(np.uint32)
    fused = np.array([(int(w) >> 5) == 0 and not live(int(w) & 31, 1, caustic) for w in wc])
    assert 50 < fused.sum() < fused.size
    assert (wf[fused] == 0).all() and (wf[~fused] == wc[~fused]).all()
    keep = torch.from_numpy(~fused)
    assert torch.equal(cut.log.rays[keep], fus.log.rays[keep])
    for k in range(cut.log.K):
        has = keep & (((cut.log.flags >> (5 * k)) & 4) != 0)
        a, b = cut.log.verts[has, k].view(torch.int32), fus.log.verts[has, k].view(torch.int32)
        assert torch.equal(a[:, :15], b[:, :15]), k                       # first sector (eta, word 15, is not drawn for a retiring vertex)
        # ... and the whole record where the path goes on behind it (a vertex that retires its path gets its first sector only)
        goes_on = (((cut.log.flags[has] >> (5 * (k + 1))) & 4) != 0) if k + 1 < cut.log.K else torch.zeros(int(has.sum()), dtype=torch.bool)
        assert torch.equal(a[goes_on], b[goes_on]), k
    # the sums, from the UNFUSED log and the float64 oracle
    v0 = cut.log.verts[:, 0]
    act = torch.from_numpy((wc & 4) != 0)
    _, dldp, go = oracle_first_vertex_tangent(cut.ray_o, cut.ray_d, cut.ray_dx, cut.ray_dy, grad_in, spp, res, v0[:, 0:3], v0[:, 3:6], v0[:, 6:9], act)
    m = float(go.abs().max())
    assert m > 0 and float((params.cam_origin.double() - go).abs().max()) <= 1e-4 * m
    tri = v0[:, 11].contiguous().view(torch.int32).long()
    rows_on = torch.from_numpy(fused & ((wc & 1) != 0)) & act & (tri >= 0)
    dp = torch.where(dldp.abs() <= 0.1, dldp, torch.zeros_like(dldp))          # the clamp of epsm.py:932-944, per component
    want = torch.zeros((params.V, 3), dtype=torch.float64)
    table = scene.tri_table.long()
    b = [v0[:, 9].double(), v0[:, 10].double()]; b.append(1 - b[0] - b[1])
    for j in range(3):
        want.index_add_(0, table[tri[rows_on], j], dp[rows_on] * b[j][rows_on, None])
    mp = float(want.abs().max())
    if caustic:            # a manifold_caustic path with a diffuse mesh first hit goes on: no path of its fused set has rows to give
        assert int(rows_on.sum()) == 0 and float(params.pos.abs().max()) == 0
    else:
        assert int(rows_on.sum()) > 20 and mp > 0
        assert float((params.pos.double() - want).abs().max()) <= 2e-4 * mp
    assert float(params.nrm.abs().max()) == 0 and float(params.alpha.abs().max() if params.B else 0) == 0
