// epsm_trace_packet.h -- WAVE-PACKET traversal of the four-wide BVH, device only.  The product uses it for the PRIMARY rays (both
// tracer forms); for bounce and visibility rays it is an A/B build that lost (-DEPSM_WF_PACKET_BOUNCE / _SHADOW: MEASUREMENTS.md 10.10).
//
// The wavefront is pixel-major, sample-minor (common.py:320-330): the 64 lanes of a wave are the samples of one pixel (spp >= 64) or of
// a few neighbouring ones -- rays that leave one point within a fraction of a degree.  The per-lane traversal (trav_round,
// epsm_trace_core.h) still pays, per lane, for a private stack, a sort of the four children and the bookkeeping of a loop whose
// lanes sit in different phases (lane utilisation 0.63, vector issue saturated: MEASUREMENTS.md 10.8).  Here the WAVE walks the tree:
// one node at a time, the same for all lanes -- its address is wave-uniform, so the node arrives by SCALAR loads and costs no vector
// registers --, every lane tests the four boxes against its own ray, a child is entered when ANY lane hits it (ballot), the order is
// that of the first interested lane's entry distances, and the stack is one LDS column per wave.  A leaf's triangles are tested by all
// lanes.  Each lane culls with its own maxt, so a subtree is skipped as soon as no lane can still be hit in it.
// Per ray the result is the closest hit of the per-lane traversal: the same box and triangle arithmetic; among hits at EXACTLY the
// same distance the one visited last wins in both, and the visiting order differs (a shared edge hit dead on: the flag words, the
// distance and the point are the same, the triangle id may be the neighbour's).
// Counters of the primary-ray stage (2^24 rays, `tools/gpu_packet_counters.sh`): 2 112 vector + 1 518 scalar + 59 scalar-memory
// instructions per wave of 64 rays (per-lane traversal: 3 360 vector), the vector pipe busy 100 % of the time.  Tried: when every
// ray of the wave looks the same way along each axis, which slab plane is the near one is a scalar choice (an offset into the
// node) and the six min / max per box go: vector instructions -16 %, but the scalar unit -- which issues as many instructions
// per cycle as the vector pipes of a CU together -- becomes the bound: 995 -> 1 100 us.  Not kept.  Nor the slab planes of two
// children per v_pk_fma_f32 (12 packed instead of 24 plain multiply-adds per node): 995 -> 1 015 us.
#pragma once

#include "epsm_trace_core.h"

namespace epsm {

constexpr int kPacketStack = 64;            // entries per wave (depth <= 16, three pushes per level: never reached)
constexpr int kPacketMaxSteps = 1 << 20;    // termination guard (never reached)

typedef const __attribute__((address_space(4))) float *ConstF;
typedef const __attribute__((address_space(4))) int32_t *ConstI;

__device__ __forceinline__ uint32_t packet_key(float t) { return __float_as_uint(t); }      // t >= 0: ordered as integers

// All 64 lanes call it; `has_ray`: this lane carries a ray.  stack: the wave's LDS column (kPacketStack words).
// ANY_HIT: a lane is done with its first hit (visibility rays); the walk ends when every lane is.
template <bool ANY_HIT>
__device__ __forceinline__ TriHit packet_intersect(const EpsmScene &S, const Ray &r0, bool has_ray, uint32_t *stack) {
    Ray r = r0;
    const F3 inv_d = f3(fminf(fmaxf(1.f / r.d.x, -1e18f), 1e18f), fminf(fmaxf(1.f / r.d.y, -1e18f), 1e18f), fminf(fmaxf(1.f / r.d.z, -1e18f), 1e18f));
    const F3 noid = f3(-r.o.x * inv_d.x, -r.o.y * inv_d.y, -r.o.z * inv_d.z);
    TriHit best; best.hit = false; best.tri = 0; best.t = r.maxt; best.u = best.v = 0.f;
    int32_t best_e = -1;
    int sp = 0;
    int32_t cur = (S.n_nodes > 0 && __ballot(has_ray) != 0ull) ? 0 : kBvhNone;              // wave-uniform
#pragma unroll 1
    for (int guard = 0; guard < kPacketMaxSteps && cur != kBvhNone; ++guard) {
        cur = __builtin_amdgcn_readfirstlane(cur);
        if (cur >= 0) {
            ConstF nb = (ConstF) (uintptr_t) (S.bvh + cur);
            ConstI nc = (ConstI) (uintptr_t) (S.bvh + cur);
            float t[4];
            unsigned long long m[4];
            int32_t ref[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float lox = nb[c], loy = nb[4 + c], loz = nb[8 + c], hix = nb[12 + c], hiy = nb[16 + c], hiz = nb[20 + c];
                ref[c] = nc[24 + c];
                const float ax = fmaf(lox, inv_d.x, noid.x), bx = fmaf(hix, inv_d.x, noid.x);
                const float ay = fmaf(loy, inv_d.y, noid.y), by = fmaf(hiy, inv_d.y, noid.y);
                const float az = fmaf(loz, inv_d.z, noid.z), bz = fmaf(hiz, inv_d.z, noid.z);
                const float t0 = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.f));
                const float t1 = fminf(fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz)) * 1.0000004f, r.maxt);
                const bool h = has_ray & (t0 <= t1) & (ref[c] != kBvhNone);
                t[c] = h ? t0 : kInf;
                m[c] = __ballot(h);
            }
            const unsigned long long any = m[0] | m[1] | m[2] | m[3];
            if (any == 0ull) { cur = sp > 0 ? (int32_t) stack[--sp] : kBvhNone; continue; }
            // the order of the first interested lane; a child only others hit comes behind its own
            const int rep = __builtin_ctzll(any);
            uint32_t key[4]; int32_t cc[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t k = (uint32_t) __builtin_amdgcn_readlane((int) packet_key(t[c]), rep);
                key[c] = m[c] != 0ull ? (k < 0x7f000000u ? k : 0x7f000000u + (uint32_t) c) : 0xffffffffu;
                cc[c] = m[c] != 0ull ? ref[c] : kBvhNone;
            }
#define EPSM_PCX(i, j) { const bool sw = key[j] < key[i]; const uint32_t ka = sw ? key[j] : key[i], kb = sw ? key[i] : key[j]; \
                         const int32_t ca = sw ? cc[j] : cc[i], cb = sw ? cc[i] : cc[j]; key[i] = ka; key[j] = kb; cc[i] = ca; cc[j] = cb; }
            EPSM_PCX(0, 1) EPSM_PCX(2, 3) EPSM_PCX(0, 2) EPSM_PCX(1, 3) EPSM_PCX(1, 2)
#undef EPSM_PCX
            if (cc[3] != kBvhNone && sp < kPacketStack) stack[sp++] = (uint32_t) cc[3];
            if (cc[2] != kBvhNone && sp < kPacketStack) stack[sp++] = (uint32_t) cc[2];
            if (cc[1] != kBvhNone && sp < kPacketStack) stack[sp++] = (uint32_t) cc[1];
            cur = cc[0];
        } else {
            const uint32_t ref = ~(uint32_t) cur;
            const int32_t first = (int32_t) (ref >> 3), count = (int32_t) (ref & 7u);
#pragma unroll 1                             // (2 / 4 triangles in flight: no faster)
            for (int32_t e = first; e < first + count; ++e) {
                ConstF q = (ConstF) (uintptr_t) (S.tri_verts + 9 * (int64_t) e);
                const F3 p0 = f3(q[0], q[1], q[2]), p1 = f3(q[3], q[4], q[5]), p2 = f3(q[6], q[7], q[8]);
                float tt, uu, vv;
                if (has_ray && moeller_trumbore(r, p0, p1, p2, tt, uu, vv)) {
                    best.hit = true; best_e = e; best.t = tt; best.u = uu; best.v = vv;
                    r.maxt = tt;
                    if (ANY_HIT) has_ray = false;
                }
            }
            if (ANY_HIT && __ballot(has_ray) == 0ull) break;
            cur = sp > 0 ? (int32_t) stack[--sp] : kBvhNone;
        }
    }
    if (best.hit) best.tri = S.prim_index[best_e];
    return best;
}

}  // namespace epsm
