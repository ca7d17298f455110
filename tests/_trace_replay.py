"""Replay of a traced vertex log against the brute-force float64 intersector of oracle/epsm_oracle_trace.c and a numpy
restatement of the sampler seeding -- the tracer's NON-SELF oracle (rows a1, a3, a11, f1 of SURVEY.md 8): nothing here
shares code with epsm_trace_core.h.  Test infrastructure."""
import numpy as np
import torch

from oracle.binding import oracle_intersect


# ---- sampler: TEA-scrambled PCG32 streams (src/render/sampler.cpp:115-134; drjit's sample_tea_32 with 4 rounds and
#      pcg32.h's seed(1, v0, v1) / next_uint32 / next_float32), vectorised over the wavefront index in numpy
def _tea32(v0, v1, rounds=4):
    v0 = v0.astype(np.uint32).copy(); v1 = v1.astype(np.uint32).copy()
    s = np.uint32(0)
    with np.errstate(over="ignore"):
        for _ in range(rounds):
            s = np.uint32((int(s) + 0x9e3779b9) & 0xFFFFFFFF)
            v0 += ((v1 << np.uint32(4)) + np.uint32(0xa341316c)) ^ (v1 + s) ^ ((v1 >> np.uint32(5)) + np.uint32(0xc8013ea4))
            v1 += ((v0 << np.uint32(4)) + np.uint32(0xad90777d)) ^ (v0 + s) ^ ((v0 >> np.uint32(5)) + np.uint32(0x7e95761e))
    return v0, v1


class Pcg32Streams:
    MULT = np.uint64(0x5851f42d4c957f2d)

    def __init__(self, seed: int, index: np.ndarray):
        v0, v1 = _tea32(np.full(index.shape, seed & 0xFFFFFFFF, dtype=np.uint32), index.astype(np.uint32))
        self.state = np.zeros(index.shape, dtype=np.uint64)
        self.inc = (v1.astype(np.uint64) << np.uint64(1)) | np.uint64(1)
        self.next_u32()
        with np.errstate(over="ignore"):
            self.state = self.state + v0.astype(np.uint64)
        self.next_u32()

    def next_u32(self):
        old = self.state
        with np.errstate(over="ignore"):
            self.state = old * self.MULT + self.inc
        xs = (((old >> np.uint64(18)) ^ old) >> np.uint64(27)).astype(np.uint32)
        rot = (old >> np.uint64(59)).astype(np.uint32)
        return (xs >> rot) | (xs << ((~rot + np.uint32(1)) & np.uint32(31)))

    def next_f32(self):
        return ((self.next_u32() >> np.uint32(9)) | np.uint32(0x3f800000)).view(np.float32) - np.float32(1.0)


def check_film_positions(trace, seed, width):
    """common.py:320-335: wavefront index // spp -> pixel (row-major), position = pixel + next_2d().  Bit-exact."""
    n = trace.ray_d.shape[0]
    idx = np.arange(trace.path_offset, trace.path_offset + n, dtype=np.int64)
    rng = Pcg32Streams(seed, idx)
    jx, jy = rng.next_f32(), rng.next_f32()
    pix = idx // trace.spp
    px, py = (pix % width).astype(np.float32), (pix // width).astype(np.float32)
    want = np.stack([px + jx, py + jy], axis=1)
    got = trace.film_pos.cpu().numpy()
    return float((want == got).all(axis=1).mean())


def _f64(t):
    return t.detach().cpu().double()


def replay(scene, trace, K, rel_tol=2e-4):
    """Re-intersects the logged rays / vertices by brute force.  Returns a dict of agreement statistics."""
    verts = _f64(scene.positions)[scene.tri.cpu().long()]                 # (T,3,3): the float32 vertices the tracer saw
    T = verts.shape[0]
    is_emitter_tri = torch.zeros(T, dtype=torch.bool)
    for m, (lo, hi) in ((m, scene.mesh_tri_slices[m.name]) for m in scene.meshes):
        if m.emitter >= 0:
            is_emitter_tri[lo:hi] = True
    out = {}
    info, sinfo = trace.path_info, trace.scatter_info
    P = lambda k: _f64(info[k]["points"][3])
    act = lambda k: info[k]["active"].cpu() > 0
    tri = lambda k: sinfo[k - 1]["tri"].cpu().long()
    mesh = lambda k: info[k]["ismesh"].cpu() > 0

    def compare(name, o, d, sel, k_hit, skip=None, tmin=0.0):
        """closest hit of rays (o, d)[sel] must be the logged vertex k_hit"""
        ids = torch.nonzero(sel).flatten()
        if ids.numel() == 0:
            return
        r = oracle_intersect(o[ids], d[ids], verts, tmin=tmin if not torch.is_tensor(tmin) else tmin[ids],
                             skip=None if skip is None else skip[ids])
        lt, lp = tri(k_hit)[ids], P(k_hit)[ids]
        dist = (lp - o[ids]).norm(dim=1)
        m = mesh(k_hit)[ids]
        same = (r["tri"] == lt)
        # a ray through a shared edge / vertex may name either neighbour: accept when the runner-up is at the same distance
        tie = (~same) & ((r["second_t"] - r["t"]).abs() <= rel_tol * (1 + r["t"]))
        t_ok = (r["t"] - dist).abs() <= rel_tol * (1 + dist)
        b0, b1 = _f64(info[k_hit]["uv"][0])[ids], _f64(info[k_hit]["uv"][1])[ids]
        uv_ok = ((1 - r["u"] - r["v"] - b0).abs() <= 5e-3) & ((r["u"] - b1).abs() <= 5e-3)      # mesh.cpp:698-700
        out[name + "_rays"] = int(ids.numel())
        out[name + "_hit_found"] = float((r["tri"] >= 0).double().mean())
        out[name + "_same_primitive"] = float((same[m] | tie[m]).double().mean()) if bool(m.any()) else 1.0
        out[name + "_exact_primitive"] = float(same[m].double().mean()) if bool(m.any()) else 1.0
        out[name + "_t_agrees"] = float(t_ok.double().mean())
        out[name + "_uv_agrees"] = float(uv_ok[m & same].double().mean()) if bool((m & same).any()) else 1.0

    # ---- camera ray -> first vertex (epsm.py:556: scene.ray_intersect of the primary ray)
    o0, d0 = _f64(trace.ray_o), _f64(trace.ray_d)
    compare("primary", o0, d0, act(1), 1)
    miss = ~act(1)
    if bool(miss.any()):
        ids = torch.nonzero(miss).flatten()
        r = oracle_intersect(o0[ids], d0[ids], verts)
        out["primary_miss_confirmed"] = float((r["tri"] < 0).double().mean())
    # ---- vertex k -> vertex k+1 (the BSDF-sampled ray of epsm.py:734, re-derived from the two logged points)
    for k in range(1, K):
        sel = act(k) & act(k + 1)
        o = P(k)
        d = torch.nn.functional.normalize(P(k + 1) - o, dim=1)
        tmin = 1e-5 * (1 + o.norm(dim=1))
        sk = torch.where(mesh(k), tri(k), torch.full_like(tri(k), -1))
        compare(f"bounce{k}", o, d, sel, k + 1, skip=sk, tmin=tmin)
    # ---- vertex k -> emitter sample: an occluded sample carries weight 0 (scene.cpp:270-275 -> epsm.py:596-605)
    occl_total = occl_zero = vis_total = vis_pos = 0
    on_light = []
    for k in range(1, K + 1):
        em = info[k]["active_em"].cpu() > 0
        rec = sinfo[k - 1]["emit"].cpu()
        etri = rec[:, 0].long()
        ew = rec[:, 3].contiguous().view(torch.float32).double()
        sel = act(k) & em
        ids = torch.nonzero(sel).flatten()
        if ids.numel() == 0:
            continue
        o, lp = P(k)[ids], _f64(info[k]["light"])[ids]
        dvec = lp - o
        dist = dvec.norm(dim=1)
        sk = torch.where(mesh(k), tri(k), torch.full_like(tri(k), -1))[ids]
        # occluders strictly between the two points: the emitter's own surface at the far end does not count
        r = oracle_intersect(o, dvec / dist[:, None], verts, tmin=1e-4 * (1 + dist), tmax=dist * (1 - 1e-3), skip=sk, any_hit=True)
        occluded = r["tri"] >= 0
        occl_total += int(occluded.sum()); occl_zero += int((occluded & (ew[ids] == 0)).sum())
        vis_total += int((~occluded).sum()); vis_pos += int(((~occluded) & (ew[ids] > 0)).sum())
        area = etri[ids] >= 0
        if bool(area.any()):
            e = etri[ids][area]
            assert bool((e < T).all()) and bool(is_emitter_tri[e].all()), "emitter record names a non-emitting triangle"
            b0 = rec[ids][area, 1].contiguous().view(torch.float32).double()
            b1 = rec[ids][area, 2].contiguous().view(torch.float32).double()
            q = verts[e]
            rebuilt = q[:, 0] * b0[:, None] + q[:, 1] * b1[:, None] + q[:, 2] * (1 - b0 - b1)[:, None]
            on_light.append(((rebuilt - lp[area]).norm(dim=1) <= 1e-4 * (1 + lp[area].norm(dim=1))).double())
    out["shadow_rays"] = occl_total + vis_total
    out["occluded_have_zero_weight"] = occl_zero / max(1, occl_total)
    out["occluded_share"] = occl_total / max(1, occl_total + vis_total)
    out["visible_have_weight"] = vis_pos / max(1, vis_total)
    if on_light:
        out["emitter_point_rebuilt"] = float(torch.cat(on_light).mean())
    return out
