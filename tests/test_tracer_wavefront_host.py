"""The wavefront form of the tracer (epsm_trace_wavefront.h: queues of live paths, extend / shade / shadow stages,
finish pass) against the one-launch form, both on the host build of tests/host_harness: every output array must be
IDENTICAL, bit for bit -- the per-path arithmetic is the same code, only the order of visiting paths differs."""
import numpy as np
import pytest
import torch

from _scenes import floor_and_light, on_host, quad, sensor
from epsm_mitsuba3_amd import scene as S


def _rich_scene(res=12, spp=8, point_light=False, occluder=False, device="cpu"):
    """Rough-conductor sphere-ish blob + glass slab + diffuse floor under an area light (and a point light): every
    BSDF branch, emitter hits after delta / smooth bounces, shadow rays that are blocked and that are not."""
    fv, ff = quad(0.0, 3.0, up=True)
    lv, lf = quad(3.0, 0.4, up=False)
    gv, gf = quad(0.8, 0.7, up=True)
    g2v, g2f = quad(0.6, 0.7, up=False)
    pv, pf = quad(0.3, 0.5, up=True)
    pv = (S.rotate([1, 0, 0], 12.0)[:3, :3] @ pv.T).T + np.array([0.9, 0.2, 0.0])
    glass = {"type": "dielectric", "int_ior": 1.5, "ext_ior": 1.0}
    d = {"type": "scene", "cam": sensor([0.3, -3.0, 2.2], [0, 0, 0.3], up=(0, 0, 1), res=res, spp=spp, rfilter="gaussian"),
         "floor": {"type": "mesh", "vertices": fv, "faces": ff, "face_normals": True,
                   "bsdf": {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.7, 0.5, 0.3]}}},
         "top": {"type": "mesh", "vertices": gv - np.array([0.8, 0, 0]), "faces": gf, "face_normals": True, "bsdf": glass},
         "bottom": {"type": "mesh", "vertices": g2v - np.array([0.8, 0, 0]), "faces": g2f, "face_normals": True, "bsdf": glass},
         "plate": {"type": "mesh", "vertices": pv, "faces": pf,
                   "bsdf": {"type": "roughconductor", "material": "Al", "distribution": "ggx", "alpha": 0.15}},
         "light": {"type": "mesh", "vertices": lv, "faces": lf, "face_normals": True,
                   "emitter": {"type": "area", "radiance": {"type": "rgb", "value": 25.0}}}}
    if point_light:
        d["pl"] = {"type": "point", "position": [1.5, -1.0, 2.5], "intensity": {"type": "rgb", "value": 8.0}}
    if occluder:
        ov, of = quad(1.5, 0.5, up=True)
        d["occ"] = {"type": "mesh", "vertices": ov, "faces": of, "face_normals": True,
                    "bsdf": {"type": "twosided", "bsdf": {"type": "diffuse"}}}
    sc = S.Scene.from_dict(d, device=device)
    if str(device) == "cpu":
        on_host(sc)
    sc.attach("plate", positions=True, normals=True)
    sc.attach_alpha("plate.bsdf")
    return sc


def _all_arrays(tr):
    out = {"ray_o": tr.ray_o, "ray_d": tr.ray_d, "ray_dx": tr.ray_dx, "ray_dy": tr.ray_dy, "film_pos": tr.film_pos,
           "radiance": tr.radiance, "valid": tr.valid}
    for k, rec in enumerate(tr.path_info[1:]):
        for name, v in rec.items():
            if isinstance(v, torch.Tensor):
                out[f"v{k}.{name}"] = v
            elif isinstance(v, (list, tuple)):
                for j, x in enumerate(v):
                    out[f"v{k}.{name}{j}"] = x
    for k, rec in enumerate(tr.scatter_info):
        for name, v in rec.items():
            if v is not None and name != "table":          # (the scene's triangle table is not a per-path array)
                out[f"s{k}.{name}"] = v
    return out


def _same(a, b):
    x, y = _all_arrays(a), _all_arrays(b)
    assert x.keys() == y.keys()
    for name in x:
        xa, ya = x[name].contiguous().view(torch.uint8), y[name].contiguous().view(torch.uint8)
        assert torch.equal(xa, ya), name                        # bit for bit (NaN-safe)


@pytest.mark.parametrize("max_depth,K,point_light,occluder", [
    (6, 5, False, False), (4, 3, True, False), (3, 2, False, True), (2, 2, True, True), (1, 1, False, False), (5, 0, True, False)])
def test_wavefront_equals_one_launch(max_depth, K, point_light, occluder):
    res, spp = 12, 8
    sc = _rich_scene(res, spp, point_light, occluder)
    n = res * res * spp
    sc.tracer = "mega"
    a = sc._trace(0, seed=5, spp=spp, max_depth=max_depth, K=K, lo=0, hi=n)
    sc.tracer = "wavefront"
    b = sc._trace(0, seed=5, spp=spp, max_depth=max_depth, K=K, lo=0, hi=n)
    _same(a, b)
    if K >= 2 and max_depth >= 3:
        v1, v2 = a.path_info[1], a.path_info[2]
        assert 0 < int((v2["active"] > 0).sum()) < int((v1["active"] > 0).sum()) <= n      # paths die on the way: compaction is exercised
    if K > 0 and 2 <= max_depth <= 3:                              # (max_depth 1: no emitter sampling at all)
        sh = a.scatter_info[0]["shadow"]
        assert sh is not None and int((sh[:, 0] != -1).sum()) > 0                           # occluder records exist


def test_wavefront_on_a_sub_range_and_ragged_tile():
    """path_offset > 0 and a path count that is no multiple of anything."""
    sc = _rich_scene(10, 4)
    lo, hi = 37, 10 * 10 * 4 - 13
    sc.tracer = "mega"
    a = sc._trace(0, seed=9, spp=4, max_depth=4, K=4, lo=lo, hi=hi)
    sc.tracer = "wavefront"
    b = sc._trace(0, seed=9, spp=4, max_depth=4, K=4, lo=lo, hi=hi)
    _same(a, b)


def test_primal_image_is_the_same():
    sc = _rich_scene(16, 8, point_light=True)
    sc.tracer = "mega"
    a = sc.render_primal(sensor=0, seed=1, spp=8, max_depth=5)
    sc.tracer = "wavefront"
    b = sc.render_primal(sensor=0, seed=1, spp=8, max_depth=5)
    assert torch.equal(a, b) and float(a.max()) > 0


def test_auto_picks_the_wavefront_for_large_scenes():
    sc = floor_and_light()
    assert sc.tracer == "auto" and not sc.use_wavefront()
    sc.WAVEFRONT_MIN_TRIANGLES = 2
    assert sc.use_wavefront()
    # ... for launches of more than 2^20 paths: below, the one launch wins (the reference's own backward size is 2^19)
    assert not sc.use_wavefront(1 << 19) and not sc.use_wavefront(1 << 20) and sc.use_wavefront((1 << 20) + 1)
    sc.tracer = "wavefront"
    assert sc.use_wavefront(1)
    sc.tracer = "nope"
    with pytest.raises(ValueError):
        sc.use_wavefront()


@pytest.mark.parametrize("tracer", ["mega", "wavefront"])
def test_sparse_log_keeps_everything_a_live_vertex_has(tracer):
    """EPSM_TRACE_SPARSE_LOG: the mask fields are written for every (path, bounce) and equal the dense log's; every
    other array equals the dense log wherever the vertex is live (what the gradient kernels read)."""
    sc = _rich_scene(12, 8, point_light=True)
    sc.tracer = tracer
    n = 12 * 12 * 8
    a = sc._trace(0, seed=5, spp=8, max_depth=5, K=4, lo=0, hi=n)
    b = sc._trace(0, seed=5, spp=8, max_depth=5, K=4, lo=0, hi=n, sparse_log=True)
    for name in ("ray_o", "ray_d", "ray_dx", "ray_dy", "film_pos", "radiance", "valid"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    dead_total = 0
    for k in range(1, 5):
        ra, rb = a.path_info[k], b.path_info[k]
        for name in ("active", "active_em", "ismesh", "bsdf"):
            assert torch.equal(ra[name], rb[name]), (k, name)
        live = ra["active"] > 0
        dead_total += int((~live).sum())
        xa, xb = _all_arrays(a), _all_arrays(b)
        for name in xa:
            if name.startswith(f"v{k - 1}.") or name.startswith(f"s{k - 1}."):
                u, v = xa[name][live], xb[name][live]
                assert torch.equal(u.contiguous().view(torch.uint8), v.contiguous().view(torch.uint8)), (k, name)
    assert dead_total > n                                           # there were dead bounces to skip


@pytest.mark.parametrize("n", [1, 65, 257])
def test_wavefront_tiny_wavefronts(n):
    sc = _rich_scene(8, 8)
    sc.tracer = "mega"
    a = sc._trace(0, seed=3, spp=8, max_depth=4, K=3, lo=5, hi=5 + n)
    sc.tracer = "wavefront"
    b = sc._trace(0, seed=3, spp=8, max_depth=4, K=3, lo=5, hi=5 + n)
    _same(a, b)


def test_wavefront_on_an_empty_scene():
    """No triangles, no emitters: every path misses; both forms log the same zeros."""
    d = {"type": "scene", "cam": sensor([1.0, 2.0, 3.0], [1.0, 2.0, -5.0], up=(0, 1, 0), res=8, spp=2)}
    sc = on_host(S.Scene.from_dict(d, device="cpu"))
    sc.tracer = "mega"
    a = sc._trace(0, seed=1, spp=2, max_depth=3, K=2, lo=0, hi=128)
    sc.tracer = "wavefront"
    b = sc._trace(0, seed=1, spp=2, max_depth=3, K=2, lo=0, hi=128)
    _same(a, b)
    assert not bool(a.valid.any()) and float(a.radiance.abs().max()) == 0


def test_wavefront_tiles_follow_the_sharding():
    """The wavefront form runs in tiles as large as the sharding allows: one rank takes the whole wavefront (up to
    2^22 paths), two ranks half each; the tiles of all ranks cover every path exactly once."""
    sc = _rich_scene(8, 8)
    sc.tile_paths = 64
    sc.tracer = "wavefront"
    n_total = 8 * 8 * 8
    one = sc.trace_paths(sensor=0, seed=1, spp=8, max_depth=3, rank=0, world_size=1)
    assert len(one) == 1 and one[0].ray_o.shape[0] == n_total
    parts = [sc.trace_paths(sensor=0, seed=1, spp=8, max_depth=3, rank=r, world_size=2) for r in (0, 1)]
    assert [len(p) for p in parts] == [1, 1]
    spans = sorted((t.path_offset, t.path_offset + t.ray_o.shape[0]) for p in parts for t in p)
    assert spans == [(0, n_total // 2), (n_total // 2, n_total)]
    both = torch.cat([t.radiance for t in sorted((t for p in parts for t in p), key=lambda t: t.path_offset)])
    assert torch.equal(both, one[0].radiance)
    sc.tracer = "mega"                                   # the one-launch form keeps the sharding unit
    assert len(sc.trace_paths(sensor=0, seed=1, spp=8, max_depth=3)) == n_total // 64


@pytest.mark.parametrize("layout", ["interleaved", "dense"])
@pytest.mark.parametrize("tracer", ["mega", "wavefront"])
@pytest.mark.parametrize("max_depth,K,occluder", [(5, 4, False), (3, 3, True), (2, 1, True)])
def test_packed_log_is_the_per_field_log_in_another_layout(tracer, max_depth, K, occluder, layout):
    """EPSM_TRACE_PACKED_LOG: the tracer writes the backward kernel's native layout (one 128-byte record per path
    vertex, rays (N,12), one flag word per path: include/epsm.h EpsmPackedLog).  Bit for bit the per-field log repacked
    by PackedLog.from_trace wherever a vertex is live; the flag words agree everywhere.  Both placements of the rays and the
    records (ABI v7: one interleaved block of K + 1 cache lines per path -- rays at word 0, records from word 16 -- or two
    dense arrays): the tracer writes through the strides of EpsmRecordOut, nothing lands outside its words."""
    from epsm_mitsuba3_amd.records import PackedLog
    res, spp = 12, 8
    sc = _rich_scene(res, spp, point_light=True, occluder=occluder)
    sc.tracer = tracer
    sc.log_layout = layout
    n = res * res * spp
    a = sc._trace(0, seed=5, spp=spp, max_depth=max_depth, K=K, lo=0, hi=n)
    b = sc._trace_packed(0, seed=5, spp=spp, max_depth=max_depth, K=K, lo=0, hi=n)
    want = PackedLog.from_trace(a, layout=layout)
    got = b.log
    assert got.layout == want.layout == layout
    if layout == "interleaved":
        assert (got.ray_stride, got.path_stride) == (32 * (K + 1), 32 * (K + 1))
        assert got.verts.data_ptr() - got.rays.data_ptr() == 64
    else:
        assert (got.ray_stride, got.path_stride) == (12, 32 * K) and got.rays.is_contiguous() and got.verts.is_contiguous()
    assert torch.equal(want.rays.view(torch.int32), got.rays.view(torch.int32))
    assert torch.equal(want.flags, got.flags)
    assert torch.equal(a.radiance, b.radiance) and torch.equal(a.film_pos, b.film_pos) and torch.equal(a.valid, b.valid)
    live_total = 0
    for k in range(K):
        live = ((got.flags >> (5 * k)) & 4) != 0
        live_total += int(live.sum())
        assert torch.equal(want.verts[live, k].view(torch.int32), got.verts[live, k].view(torch.int32)), k
    assert live_total > n // 2
    if max_depth <= 3:
        assert torch.equal(want.shadow, got.shadow)
    else:
        assert got.shadow is None


def test_packed_record_words_are_where_the_header_says():
    """include/epsm.h, EpsmPackedLog (ABI v5): p0 p1 p2 at 0..8, b0 b1 at 9, 10, the triangle id at 11, n0 at 12..14, eta at 15 -- the
    first 64-byte sector --, n1 n2 at 16..21, light at 22, 23 and 28, the emitter sample at 24..27, d hf / d alpha at 29..31; checked
    word by word against the per-field log of the same trace."""
    res, spp, K = 12, 8, 3
    sc = _rich_scene(res, spp, point_light=False, occluder=True)
    sc.tracer = "mega"
    n = res * res * spp
    a = sc._trace(0, seed=5, spp=spp, max_depth=4, K=K, lo=0, hi=n)
    b = sc._trace_packed(0, seed=5, spp=spp, max_depth=4, K=K, lo=0, hi=n).log
    w = b.verts
    iw = w.view(torch.int32)
    seen = 0
    for k in range(K):
        r, s = a.path_info[k + 1], a.scatter_info[k]
        live = r["active"].bool()
        seen += int(live.sum())
        f = lambda t: t[live].float()
        for j in range(3):
            assert torch.equal(w[live, k, 3 * j: 3 * j + 3], f(r["points"][j]))
        assert torch.equal(w[live, k, 9], f(r["uv"][0])) and torch.equal(w[live, k, 10], f(r["uv"][1]))
        assert torch.equal(iw[live, k, 11], s["tri"][live].to(torch.int32))
        assert torch.equal(w[live, k, 12:15], f(r["normals"][0])) and torch.equal(w[live, k, 15], f(r["eta"]))
        assert torch.equal(w[live, k, 16:19], f(r["normals"][1])) and torch.equal(w[live, k, 19:22], f(r["normals"][2]))
        light = f(r["light"])
        assert torch.equal(w[live, k, 22:24], light[:, :2]) and torch.equal(w[live, k, 28], light[:, 2])
        assert torch.equal(iw[live, k, 24:28], s["emit"][live].to(torch.int32))
        if s.get("aux") is not None:
            assert torch.equal(iw[live, k, 29:32], s["aux"][live][:, 1:4].to(torch.int32))
    assert seen > n
