// epsm_trace_quad.h -- CHILD-PARALLEL traversal of the four-wide BVH (round 5): FOUR LANES PER RAY, sixteen rays
// per wave.  Lane c of a quad tests child c's box of the current node (one slab test instead of four), the four (distance, child)
// pairs are sorted ACROSS the quad with the same five compare-exchanges as the per-lane traversal (three DPP stages), the nearest
// is followed and the others pushed farthest first onto the ray's stack (one LDS column per quad, 64 entries: no overflow area);
// in a leaf lane c tests triangle first + c (+ 4 for the leaf's second group), the quad's closest hit by two DPP stages.  Same
// nodes, same order, same arithmetic per box and triangle as trav_round (epsm_trace_core.h) -- and the same answer on ties: among
// equal distances the triangle LATER in leaf order wins, as the sequential "t <= maxt replaces" does.
// What it changes: a node step costs one box test per lane instead of four (all four lanes busy by construction), and a wave waits
// for the slowest of 16 rays instead of 64.  Device only.
#pragma once

#include "epsm_trace_core.h"

namespace epsm {

template <int CTRL> __device__ __forceinline__ float quad_f(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int CTRL> __device__ __forceinline__ int32_t quad_i(int32_t v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true); }
constexpr int kQuadSwap1 = 0xB1, kQuadSwap2 = 0x4E, kQuadMid = 0xD8, kQuadBcast0 = 0x00;      // quad_perm [1,0,3,2] [2,3,0,1] [0,2,1,3] [0,0,0,0]

// compare-exchange of (t, c) with the partner the permutation names; `lower`: this lane keeps the smaller one (strictly: equal
// distances stay where they are, as EPSM_CX in trav_round)
template <int CTRL> __device__ __forceinline__ void quad_cx(float &t, int32_t &c, bool lower) {
    const float tp = quad_f<CTRL>(t);
    const int32_t cq = quad_i<CTRL>(c);
    const bool take = lower ? (tp < t) : (t < tp);
    t = take ? tp : t; c = take ? cq : c;
}

constexpr int kQuadStack = 64;              // entries per ray (kBvhStack = 48 can never be exceeded: depth <= 16, three pushes per level)
constexpr int kQuadMaxSteps = 1 << 16;      // a wave leaves the loop after this many rounds whatever its state (never reached: termination guard)

// All 64 lanes call it; `has_ray` is quad-uniform.  stack: the quad's LDS column, entry k at stack[k * stride].
template <bool ANY_HIT>
__device__ __forceinline__ TriHit quad_intersect(const EpsmScene &S, const Ray &r0, bool has_ray, uint32_t *stack_, int stride, int lane) {
    typedef __attribute__((address_space(3))) uint32_t LdsWord;
    LdsWord *stack = (LdsWord *) stack_;
    const int c = lane & 3;
    Ray r = r0;
    const F3 inv_d = f3(fminf(fmaxf(1.f / r.d.x, -1e18f), 1e18f), fminf(fmaxf(1.f / r.d.y, -1e18f), 1e18f), fminf(fmaxf(1.f / r.d.z, -1e18f), 1e18f));
    const F3 noid = f3(-r.o.x * inv_d.x, -r.o.y * inv_d.y, -r.o.z * inv_d.z);
    TriHit best; best.hit = false; best.tri = 0; best.t = r.maxt; best.u = best.v = 0.f;
    int32_t best_e = -1;
    int32_t cur = (has_ray && S.n_nodes > 0) ? 0 : kBvhNone;
    int sp = 0;
    const int qshift = lane & ~3;
#pragma unroll 1
    for (int guard = 0; guard < kQuadMaxSteps && __ballot(cur != kBvhNone) != 0ull; ++guard) {
        // ---- descend: node steps until no quad of the wave stands on an inner node
#pragma unroll 1
        for (int g2 = 0; g2 < kQuadMaxSteps; ++g2) {
            const bool inner = cur >= 0 && cur != kBvhNone;
            if (__ballot(inner) == 0ull) break;
            if (inner) {
                const float *nb = (const float *) (S.bvh + cur);
                const float lox = nb[c], loy = nb[4 + c], loz = nb[8 + c], hix = nb[12 + c], hiy = nb[16 + c], hiz = nb[20 + c];
                const int32_t ref = ((const int32_t *) nb)[24 + c];
                const float ax = fmaf(lox, inv_d.x, noid.x), bx = fmaf(hix, inv_d.x, noid.x);
                const float ay = fmaf(loy, inv_d.y, noid.y), by = fmaf(hiy, inv_d.y, noid.y);
                const float az = fmaf(loz, inv_d.z, noid.z), bz = fmaf(hiz, inv_d.z, noid.z);
                const float t0 = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.f));
                const float t1 = fminf(fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz)) * 1.0000004f, r.maxt);
                const bool h = (t0 <= t1) & (ref != kBvhNone);
                float t = h ? t0 : kInf;
                int32_t cc = h ? ref : kBvhNone;
                quad_cx<kQuadSwap1>(t, cc, (c & 1) == 0);            // (0,1) (2,3)
                quad_cx<kQuadSwap2>(t, cc, (c & 2) == 0);            // (0,2) (1,3)
                quad_cx<kQuadMid>(t, cc, c == 1);                    // (1,2); lanes 0 and 3 are their own partners: nothing moves
                // lane k now holds the k-th nearest child; the hit ones are lanes 0 .. nh - 1
                const unsigned long long m = __ballot(cc != kBvhNone);
                const int nh = __popc((unsigned) ((m >> qshift) & 0xFull));
                const int32_t nearest = quad_i<kQuadBcast0>(cc);
                if (c >= 1 && c < nh && sp + (nh - 1 - c) < kQuadStack) stack[(sp + (nh - 1 - c)) * stride] = (uint32_t) cc;
                if (nh > 0) { cur = nearest; sp += nh - 1; if (sp > kQuadStack) sp = kQuadStack; }
                else cur = sp > 0 ? (int32_t) stack[(--sp) * stride] : kBvhNone;
            }
        }
        // ---- the leaves the quads stand on
        if (cur != kBvhNone && cur < 0) {
            const uint32_t ref = ~(uint32_t) cur;
            const int32_t first = (int32_t) (ref >> 3), count = (int32_t) (ref & 7u);
            bool done = false;
#pragma unroll 1
            for (int32_t base = 0; base < count && !done; base += 4) {
                const int32_t e = first + base + c;
                float t = kInf, u = 0.f, v = 0.f;
                bool hit = false;
                if (base + c < count) {
                    const float *q = S.tri_verts + 9 * (int64_t) e;
                    float tt, uu, vv;
                    if (moeller_trumbore(r, ld3(q), ld3(q + 3), ld3(q + 6), tt, uu, vv)) { hit = true; t = tt; u = uu; v = vv; }
                }
                int32_t ee = hit ? e : -1;
                // the quad's winner: smallest t, among equal ones the LATER triangle (the sequential test's `t <= maxt` replaces)
#define EPSM_QUAD_MIN(CTRL) { const float tp = quad_f<CTRL>(t), up = quad_f<CTRL>(u), vp = quad_f<CTRL>(v); const int32_t ep = quad_i<CTRL>(ee); \
                              const bool take = (ep >= 0) && (ee < 0 || tp < t || (tp == t && ep > ee)); \
                              t = take ? tp : t; u = take ? up : u; v = take ? vp : v; ee = take ? ep : ee; }
                EPSM_QUAD_MIN(kQuadSwap1) EPSM_QUAD_MIN(kQuadSwap2)
#undef EPSM_QUAD_MIN
                if (ee >= 0) {
                    best.hit = true; best_e = ee; best.t = t; best.u = u; best.v = v;
                    r.maxt = t;
                    if (ANY_HIT) done = true;
                }
            }
            if (ANY_HIT && best.hit) cur = kBvhNone;
            else cur = sp > 0 ? (int32_t) stack[(--sp) * stride] : kBvhNone;
        }
    }
    if (best.hit) best.tri = S.prim_index[best_e];
    return best;
}

}  // namespace epsm
