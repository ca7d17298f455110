// epsm_trace_wavefront.h -- the tracer as a wavefront of stages with compaction between bounces
// (include/epsm_trace.h: epsm_trace_paths_wavefront).
//
// The reference runs sample_path as a Python-unrolled wavefront (one Dr.Jit kernel per bounce over ALL lanes,
// masked, epsm.py:551).  epsm_trace_paths keeps a lane on its path through every bounce of one launch; on a scene
// with many triangles that kernel is bound by divergence: dead paths stay in their wave, lanes without an emitter
// sample idle through the shadow-ray traversal of their neighbours, and the 148 registers of the whole path cap
// the occupancy of what is a pointer-chasing loop.  Here a bounce is three small kernels over QUEUES of live paths:
//
//   extend   closest hit of the path's ray            (traversal only: few registers, many waves per SIMD)
//   shade    path_bounce() of epsm_trace_core.h: surface interaction, emission + MIS, emitter sample, BSDF sample,
//            vertex log, Russian roulette; appends the path to the next bounce's queue if it lives on and to the
//            shadow queue if its emitter sample needs a visibility ray (wave-level ballot + prefix count, one
//            atomic per wave)
//   shadow   any-hit of the queued visibility rays; adds the direct light or zeroes the logged emitter weight;
//            the occluder record of the first vertex (epsm.py:609-620) when the integrator asks for it
//
// and a last pass (`finish`) writes radiance / valid and the inactive-zero records of the bounces a path never
// reached (the reference logs masked lanes as zeros).  Per-path results are identical to epsm_trace_paths: the
// arithmetic is the same code (path_begin / path_bounce / path_end), only the order in which paths are visited
// differs.  State between the stages lives in a caller-provided workspace, ten 16-byte words per path (+ 3 queue
// entries and the 64-byte overflow area of its traversal stack).
#pragma once

#include "epsm_trace_core.h"

namespace epsm {

struct alignas(16) W4 { uint32_t x, y, z, w; };
EPSM_HD W4 pack4(F3 v, float w) { W4 o; o.x = f2u(v.x); o.y = f2u(v.y); o.z = f2u(v.z); o.w = f2u(w); return o; }
EPSM_HD W4 pack4u(F3 v, uint32_t w) { W4 o; o.x = f2u(v.x); o.y = f2u(v.y); o.z = f2u(v.z); o.w = w; return o; }
EPSM_HD F3 xyz(const W4 &q) { return f3(u2f(q.x), u2f(q.y), u2f(q.z)); }

constexpr int kWfArrays = 10;                 // 16-byte words per path in the workspace
constexpr int kWfMaxDepth = 6;                // epsm.py:549
constexpr int kWfFirstHitSlots = 256;         // EPSM_TRACE_FUSE_FIRST_HIT: partial sums of grad_d, [slot][4] floats behind the counters
constexpr int kWfCounters = 64;               // uint32 header: [b] = paths alive into bounce b, [8 + b] = shadow rays of bounce b
constexpr uint32_t kWfOccluder = 1u;          // sh_d.w: the shadow stage also owes the occluder record

struct WfState {                              // device pointers into the workspace
    uint32_t *counters;
    W4 *ray_o;        // ray.o, ray.maxt
    W4 *ray_d;        // ray.d, eta
    W4 *hit;          // tri (kNoIndex: miss), t, u, v
    W4 *beta;         // beta, prev_bsdf_pdf
    W4 *L;            // L, depth | prev_bsdf_delta << 8 | bounces done << 16
    W4 *prev_p;       // prev_p, flag word of the vertices logged so far (PathState::gword)
    W4 *rng;          // state lo, hi, inc lo, hi
    W4 *sh_o;         // visibility ray o, maxt
    W4 *sh_d;         // visibility ray d, kWfOccluder
    W4 *sh_L;         // Lr_dir of the pending emitter sample, -
    uint32_t *queue[2];   // path ids alive into bounce b: queue[b & 1]
    uint32_t *shadow_queue;
    uint32_t *stack_ovf;  // (kWfStackOvf, N): traversal-stack entries beyond the kWfStackLds a path keeps in LDS
    // compaction without atomics (one hot queue counter serialises ~10 ns per wave: 0.6 ms per queue and bounce at
    // 4 M paths): the shade stage leaves a flag per queue slot and two counts per 256-slot chunk, a one-workgroup
    // scan turns the counts into offsets, a compaction pass writes the two queues -- in path order
    uint8_t *flags;           // per slot of the current queue: kWfAlive | kWfShadow
    uint32_t *chunk_counts;   // (2, chunks): alive, shadow per chunk; exclusive offsets after the scan
    uint32_t *group_counts;   // (2, groups): the same per GROUP of kWfGroup chunks (summed by the shade stage; the scan zeroes them again)
    uint32_t *group_offsets;  // (2, groups): exclusive offsets of the groups (scan, first level)
    float *fh_partial;        // (kWfFirstHitSlots, 4): partial sums of grad_d (EPSM_TRACE_FUSE_FIRST_HIT), zeroed with the counters
    int64_t chunks;           // ceil(N / kWfChunk)
    int64_t groups;           // ceil(chunks / kWfGroup)
    int64_t N;
};
#ifndef EPSM_WF_CHUNK
#define EPSM_WF_CHUNK 256
#endif
constexpr int kWfChunk = EPSM_WF_CHUNK;      // slots per chunk = lanes of a shade / compaction workgroup (64 / 128 / 512: the backward
                                              // trace of 2^24 paths 3.76 / 3.58 / 3.59 ms against 3.47)
constexpr int kWfGroup = 64;                  // chunks per group: one wave scans a group
constexpr uint8_t kWfAlive = 1, kWfShadow = 2;
constexpr int kWfStackLds = 16, kWfStackOvf = kBvhStack - kWfStackLds;
// path i's traversal stack: `lds` = its LDS column (entry k at lds[k * stride])
EPSM_HD BvhStack wf_stack(const WfState &W, int64_t i, uint32_t *lds, int stride) {
    BvhStack st{lds, stride};
    st.cap = kWfStackLds; st.ovf = W.stack_ovf + i; st.ovf_stride = W.N;
    return st;
}
EPSM_HD size_t wf_align(size_t x) { return (x + 255) & ~(size_t) 255; }
EPSM_HD size_t wf_workspace_bytes(int64_t N) {
    const size_t chunks = (size_t) ((N + kWfChunk - 1) / kWfChunk), groups = (chunks + kWfGroup - 1) / kWfGroup;
    return wf_align(kWfCounters * 4) + wf_align(kWfFirstHitSlots * 16) + 2 * wf_align(groups * 8) + (size_t) kWfArrays * wf_align((size_t) N * 16) + 3 * wf_align((size_t) N * 4) +
           wf_align((size_t) N * 4 * kWfStackOvf) + wf_align((size_t) N) + wf_align(chunks * 8);
}
// what a trace zeroes before its first stage: the counters and, behind them, the group counts
EPSM_HD size_t wf_zeroed_bytes(int64_t N) {
    const size_t chunks = (size_t) ((N + kWfChunk - 1) / kWfChunk), groups = (chunks + kWfGroup - 1) / kWfGroup;
    return wf_align(kWfCounters * 4) + wf_align(kWfFirstHitSlots * 16) + wf_align(groups * 8);
}
EPSM_HD WfState wf_carve(void *workspace, int64_t N) {
    char *p = (char *) workspace;
    WfState W;
    W.counters = (uint32_t *) p; p += wf_align(kWfCounters * 4);
    W.fh_partial = (float *) p; p += wf_align(kWfFirstHitSlots * 16);
    W.chunks = (N + kWfChunk - 1) / kWfChunk;
    W.groups = (W.chunks + kWfGroup - 1) / kWfGroup;
    W.group_counts = (uint32_t *) p; p += wf_align((size_t) W.groups * 8);
    W.group_offsets = (uint32_t *) p; p += wf_align((size_t) W.groups * 8);
    W4 **arr[kWfArrays] = {&W.ray_o, &W.ray_d, &W.hit, &W.beta, &W.L, &W.prev_p, &W.rng, &W.sh_o, &W.sh_d, &W.sh_L};
    for (int a = 0; a < kWfArrays; ++a) { *arr[a] = (W4 *) p; p += wf_align((size_t) N * 16); }
    W.queue[0] = (uint32_t *) p; p += wf_align((size_t) N * 4);
    W.queue[1] = (uint32_t *) p; p += wf_align((size_t) N * 4);
    W.shadow_queue = (uint32_t *) p; p += wf_align((size_t) N * 4);
    W.stack_ovf = (uint32_t *) p; p += wf_align((size_t) N * 4 * kWfStackOvf);
    W.flags = (uint8_t *) p; p += wf_align((size_t) N);
    W.chunk_counts = (uint32_t *) p;
    W.N = N;
    return W;
}

EPSM_HD void wf_store(const WfState &W, int64_t i, const PathState &s, int done) {
    W.ray_o[i] = pack4(s.ray.o, s.ray.maxt);
    W.ray_d[i] = pack4(s.ray.d, s.eta);
    W.beta[i] = pack4(s.beta, s.prev_bsdf_pdf);
    W.L[i] = pack4u(s.L, (uint32_t) s.depth | (s.prev_bsdf_delta ? 0x100u : 0u) | ((uint32_t) done << 16));
    W.prev_p[i] = pack4u(s.prev_p, s.gword);
    W4 r; r.x = (uint32_t) s.rng.state; r.y = (uint32_t) (s.rng.state >> 32); r.z = (uint32_t) s.rng.inc; r.w = (uint32_t) (s.rng.inc >> 32);
    W.rng[i] = r;
}
EPSM_HD PathState wf_load(const WfState &W, int64_t i) {
    PathState s;
    const W4 o = W.ray_o[i], d = W.ray_d[i], b = W.beta[i], l = W.L[i], r = W.rng[i];
    s.ray.o = xyz(o); s.ray.maxt = u2f(o.w);
    s.ray.d = xyz(d); s.eta = u2f(d.w);
    s.beta = xyz(b); s.prev_bsdf_pdf = u2f(b.w);
    s.L = xyz(l); s.depth = (int) (l.w & 0xFFu); s.prev_bsdf_delta = (l.w & 0x100u) != 0;
    const W4 pp = W.prev_p[i];
    s.prev_p = xyz(pp); s.gword = pp.w;
    s.rng.state = (uint64_t) r.x | ((uint64_t) r.y << 32); s.rng.inc = (uint64_t) r.z | ((uint64_t) r.w << 32);
    s.active = true;                                                     // only live paths are queued
    return s;
}

// ---- (no generate stage: every path is alive into bounce 0, the queue of bounce 0 is the identity, and both stages of
//      bounce 0 derive the primary ray from the path index -- sample_rays, ~100 instructions -- instead of one kernel
//      writing 96 bytes of state per path for the next two to read back: 0.45 ms + 0.2 ms per 2^24 paths)

// ---- stage: extend.  Closest hit of path i's ray, as a resumable job (begin, traversal rounds until done).
struct WfJob {
    int64_t i;
    int phase;                     // shadow stage: 0 = visibility (any hit), 1 = occluder record (closest hit)
    Traversal T;
};
EPSM_HD void wf_extend_begin(const TraceArgs &A, const WfState &W, int64_t i, WfJob &J, int iteration) {
    Ray r;
    if (iteration == 0) {
        r = path_begin(A, i, false).ray;
    } else {
        const W4 o = W.ray_o[i], d = W.ray_d[i];
        r.o = xyz(o); r.maxt = u2f(o.w); r.d = xyz(d);
    }
    J.i = i; J.phase = 0;
    trav_begin(J.T, A.S, r);
}
// one round; true when the job is finished (its result is written)
EPSM_HD bool wf_extend_round(const TraceArgs &A, const WfState &W, WfJob &J, uint32_t *lds, int stride) {
    if (!trav_done(J.T)) trav_round<false>(J.T, A.S, wf_stack(W, J.i, lds, stride));
    if (!trav_done(J.T)) return false;
    const TriHit th = trav_result(J.T, A.S);
    W4 h; h.x = th.hit ? th.tri : kNoIndex; h.y = f2u(th.t); h.z = f2u(th.u); h.w = f2u(th.v);
    W.hit[J.i] = h;
    return true;
}
EPSM_HD void wf_extend(const TraceArgs &A, const WfState &W, int64_t i, uint32_t *lds, int stride, int iteration) {
    WfJob J;
    wf_extend_begin(A, W, i, J, iteration);
    while (!wf_extend_round(A, W, J, lds, stride)) {}
}

// Visibility policy of the shade stage: nothing is traced, the questions are written down for the shadow stage.
struct DeferredVis {
    bool pending, want_occluder;
    Ray sr;
    F3 Lr;
    // EPSM_TRACE_FUSE_FIRST_HIT: `gd` in (wf_shade, bounce 0), the rows out
    F3 gd;
    FirstHitRows fh;
    bool fh_taken, fh_enabled;       // fh_enabled: the caller collects the rows (wf_shade's `out`); otherwise the vertex is logged as ever
    EPSM_HD bool first_hit(const TraceArgs &A, int64_t i, uint32_t w, const SurfHit &si, const Ray &ray) {
        if (!fh_enabled) return false;
        fh_taken = true;
        fh = first_hit_rows(A, w, si, ray, gd);
        A.rec[0].pflags[i] = 0u;                                         // no vertex: the backward kernel gives this path no lane
        return true;
    }
    EPSM_HD bool occluded(const EpsmScene &, const Ray &r) { pending = true; sr = r; return false; }   // optimistic
    EPSM_HD void direct(F3 &L, F3 Le, F3 Lr_dir) { L = L + Le; Lr = Lr_dir; }                           // + Lr_dir if visible
    EPSM_HD void occluder(const TraceArgs &A, int64_t i, const SurfHit &, const EmitterSample &, bool active_em) {
        if (A.max_depth <= 3 && active_em) want_occluder = true;         // active_em => the visibility ray is pending too
        else write_no_occluder(A.rec[0].shadow + 4 * i);
    }
};

// ---- stage: shade.  Returns through `alive` / `shadow` whether path i goes on to bounce `iteration + 1` / has a
//      visibility ray pending; the caller compacts.
// what path i gives the backward pass at its first hit (EPSM_TRACE_FUSE_FIRST_HIT; the caller sums it over the wave and adds it)
struct WfFirstHit { FirstHitRows rows; F3 gd; };
EPSM_HD void wf_shade(const TraceArgs &A, const WfState &W, int64_t i, int iteration, bool &alive, bool &shadow, WfFirstHit *out = nullptr) {
    DeferredVis vis; vis.pending = false; vis.want_occluder = false; vis.Lr = zero3<float>();
    vis.sr.o = vis.sr.d = zero3<float>(); vis.sr.maxt = 0.f;
    vis.gd = zero3<float>(); vis.fh_taken = false; vis.fh_enabled = out != nullptr; vis.fh.on = false; vis.fh.key[0] = vis.fh.key[1] = vis.fh.key[2] = kNoIndex;
    vis.fh.val[0] = vis.fh.val[1] = vis.fh.val[2] = zero3<float>();
    PathState s;
    const bool fuse = iteration == 0 && (A.flags & EPSM_TRACE_FUSE_FIRST_HIT);
    if (fuse) {
        PrimaryRay pr;
        s = path_begin(A, i, false, &pr);                               // (the rays are logged below, for the paths that keep a log)
        if (out) {                                                      // (not when the closest-hit stage has dealt with the first hit already)
            float gx, gy;
            first_hit_pixel_grad(A, i, gx, gy);
            vis.gd = (pr.dx - pr.ray.d) * gx + (pr.dy - pr.ray.d) * gy;  // epsm.py:255, as tangent_from forms it
        }
    } else {
        s = iteration == 0 ? path_begin(A, i) : wf_load(W, i);
    }
    const W4 h = W.hit[i];
    TriHit th; th.hit = h.x != kNoIndex; th.tri = th.hit ? h.x : 0u; th.t = u2f(h.y); th.u = u2f(h.z); th.v = u2f(h.w);
    path_bounce(A, i, iteration, s, th, vis);
    if (out) { out->rows = vis.fh; out->gd = vis.gd; }
    // a path the first-hit stage has dealt with leaves nothing in the log, its rays included (48 bytes x 94 % of the paths of the
    // clutter scene); the others get theirs now -- derived again rather than carried through the bounce in twelve registers
    if (fuse && !vis.fh_taken) path_begin(A, i, true);
    alive = s.active && iteration + 1 < path_max_depth(A);
    if (alive) {
        wf_store(W, i, s, iteration + 1);
    } else {                                                             // a path that ends here: only what finish / shadow read
        // (nothing when nobody asked for radiance / valid and the log is the native one: the gradient-only trace)
        if (A.radiance || A.valid || !(A.flags & EPSM_TRACE_PACKED_LOG))
            W.L[i] = pack4u(s.L, (uint32_t) s.depth | ((uint32_t) (iteration + 1) << 16));
        if (vis.want_occluder) W.prev_p[i] = pack4(s.prev_p, 0.f);
    }
    if (vis.pending) {
        W.sh_o[i] = pack4(vis.sr.o, vis.sr.maxt);
        W.sh_d[i] = pack4u(vis.sr.d, vis.want_occluder ? kWfOccluder : 0u);
        W.sh_L[i] = pack4(vis.Lr, 0.f);
    }
    shadow = vis.pending;
}

// ---- stage: shadow.  The visibility ray of path i's emitter sample at bounce `iteration` (any hit), then -- when the
//      shade stage asked for it -- the occluder record of the first vertex (closest hit); a resumable job like extend.
EPSM_HD void wf_shadow_begin(const TraceArgs &A, const WfState &W, int64_t i, WfJob &J) {
    const W4 o = W.sh_o[i], d = W.sh_d[i];
    Ray sr; sr.o = xyz(o); sr.maxt = u2f(o.w); sr.d = xyz(d);
    J.i = i; J.phase = 0;
    trav_begin(J.T, A.S, sr);
}
// ds.p of the first bounce, as logged (per-field array or packed record)
EPSM_HD F3 wf_logged_light0(const TraceArgs &A, int64_t i) {
    if (A.flags & EPSM_TRACE_PACKED_LOG) {                                 // light = words 22, 23, 28 (include/epsm.h, EpsmPackedLog)
        const float *r = packed_record(A, i, 0);
        return f3(r[22], r[23], r[28]);
    }
    return ld3(A.rec[0].light + 3 * i);
}
// the visibility ray's answer applied; true when the occluder record is owed as well
EPSM_HD bool wf_shadow_resolve(const TraceArgs &A, const WfState &W, int64_t i, int iteration, bool occluded) {
    if (occluded) {
        if (iteration < A.K_log) {                                       // Lr_dir = 0: the logged weight with it
            if (A.flags & EPSM_TRACE_PACKED_LOG) packed_record(A, i, iteration)[27] = 0.f;
            else A.rec[iteration].emit[4 * i + 3] = 0u;
        }
    } else if (A.radiance) {
        const W4 l = W.L[i];
        const F3 L = xyz(l) + xyz(W.sh_L[i]);                            // (L + Le) + Lr_dir, epsm.py:658
        W.L[i] = pack4u(L, l.w);
    }
    return (W.sh_d[i].w & kWfOccluder) != 0;
}
// iteration 0, max_depth <= 3, K_log > 0: si.p is the path's prev_p by now, ds.p was logged as the vertex's
// light point; ds.d as in sample_emitter_direction, the origin of spawn_ray(si, ds.d) is that of the visibility ray
EPSM_HD Ray wf_occluder_ray(const TraceArgs &A, const WfState &W, int64_t i, F3 &sip, F3 &esp) {
    sip = xyz(W.prev_p[i]); esp = wf_logged_light0(A, i);
    const F3 dd = esp - sip;
    const float dist = sqrtf(dot(dd, dd));
    Ray r2; r2.o = xyz(W.sh_o[i]); r2.d = dd * (1.f / dist); r2.maxt = kInf;
    return r2;
}
EPSM_HD bool wf_shadow_round(const TraceArgs &A, const WfState &W, int iteration, WfJob &J, uint32_t *lds, int stride) {
    const BvhStack st = wf_stack(W, J.i, lds, stride);
    const int64_t i = J.i;
    F3 sip, esp;
    if (J.phase == 0) {
        if (!trav_done(J.T)) trav_round<true>(J.T, A.S, st);
        if (!trav_done(J.T)) return false;
        if (!wf_shadow_resolve(A, W, i, iteration, J.T.best.hit)) return true;
        J.phase = 1;
        trav_begin(J.T, A.S, wf_occluder_ray(A, W, i, sip, esp));
        return false;
    }
    if (!trav_done(J.T)) trav_round<false>(J.T, A.S, st);
    if (!trav_done(J.T)) return false;
    const Ray r2 = wf_occluder_ray(A, W, i, sip, esp);
    write_occluder(A.S, A.rec[0].shadow + 4 * i, r2, trav_result(J.T, A.S), sip, esp);
    return true;
}
EPSM_HD void wf_shadow(const TraceArgs &A, const WfState &W, int64_t i, int iteration, uint32_t *lds, int stride) {
    WfJob J;
    wf_shadow_begin(A, W, i, J);
    while (!wf_shadow_round(A, W, iteration, J, lds, stride)) {}
}

// ---- the tail.  A stage's time has a floor of one traversal's dependent misses (~0.1 ms: every launch starts with cold
//      caches), so the bounces a few thousand paths reach cost five floors each.  Once the queue into bounce b is shorter than
//      kWfTailBelow, ONE launch carries those paths through the rest of their loop, every visibility ray answered on the spot --
//      the same path_bounce, so the same results per path.  The stages of the bounces >= b find wf_in_tail() and leave.
#ifndef EPSM_WF_TAIL_BELOW
#define EPSM_WF_TAIL_BELOW (1 << 19)
#endif
constexpr int64_t kWfTailBelow = EPSM_WF_TAIL_BELOW;
EPSM_HD bool wf_in_tail(const TraceArgs &A, int b, int64_t count) {
    return b >= 1 && !(A.flags & EPSM_TRACE_NO_TAIL) && count < kWfTailBelow;
}
EPSM_HD void wf_tail(const TraceArgs &A, const WfState &W, int64_t i, int b, uint32_t *lds, int stride) {
    PathState s = wf_load(W, i);
    const BvhStack st = wf_stack(W, i, lds, stride);
    InlineVis vis{st};
    const int max_depth = path_max_depth(A);
    int iteration = b;
    for (; iteration < max_depth && s.active; ++iteration) {
        const TriHit th = intersect<false>(A.S, s.ray, st);
        path_bounce(A, i, iteration, s, th, vis);
    }
    // as the shade stage leaves a path that ends: what finish reads
    if (A.radiance || A.valid || !(A.flags & EPSM_TRACE_PACKED_LOG))
        W.L[i] = pack4u(s.L, (uint32_t) s.depth | ((uint32_t) iteration << 16));
}

// A bounce the path never reached: what path_bounce logs for a masked lane (epsm.py:551: inactive zeros).
EPSM_HD void write_dead_record(const EpsmRecordOut &R, int64_t i) {
    const F3 z = zero3<float>();
    st3(R.p0, i, z); st3(R.p1, i, z); st3(R.p2, i, z); st3(R.p, i, z);
    st3(R.n0, i, z); st3(R.n1, i, z); st3(R.n2, i, z); st3(R.normal, i, z);
    st1(R.b0 + i, 0.f); st1(R.b1 + i, 0.f); st1(R.eta + i, 0.f);
    st3(R.hf, i, z); st3(R.light, i, z);
    st1(R.bsdf + i, 0u);
    st1(R.active + i, (uint8_t) 0); st1(R.active_em + i, (uint8_t) 0); st1(R.ismesh + i, (uint8_t) 0);
    st1(R.tri + i, kNoIndex);
    uint32_t *a = R.aux + 4 * i; st1(a, kNoIndex); st1(a + 1, 0u); st1(a + 2, 0u); st1(a + 3, 0u);
    uint32_t *e = R.emit + 4 * i; st1(e, kNoIndex); st1(e + 1, 0u); st1(e + 2, 0u); st1(e + 3, 0u);
}

// ---- stage: finish.  radiance / valid of path i and the records of the bounces it did not reach.
EPSM_HD void wf_finish(const TraceArgs &A, const WfState &W, int64_t i) {
    const W4 l = W.L[i];
    st3(A.radiance, i, xyz(l));
    if (A.valid) A.valid[i] = (l.w & 0xFFu) != 0;
    const int done = (int) (l.w >> 16);
    if (A.flags & EPSM_TRACE_PACKED_LOG) {
        // nothing: the flag word of the path says which records exist
    } else if (A.flags & EPSM_TRACE_SPARSE_LOG) {
        for (int k = done; k < A.K_log; ++k) write_dead_masks(A.rec[k], i);
    } else {
        for (int k = done; k < A.K_log; ++k) write_dead_record(A.rec[k], i);
    }
}

}  // namespace epsm
