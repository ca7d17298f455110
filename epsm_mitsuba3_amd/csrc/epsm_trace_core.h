// epsm_trace_core.h -- per-path code of the wavefront tracer (include/epsm_trace.h).
//
// Restates, for triangle meshes and the plugin set the EPSM experiments use
// (EPSM/exp/*.py: obj/ply/rectangle, diffuse, roughconductor, dielectric, twosided,
// area, point, perspective, hdrfilm + gaussian, independent), the unidirectional path
// tracer with emitter sampling + MIS of EPSMIntegrator.sample_path (epsm.py:503-742)
// including its vertex log (epsm.py:547, 648-654).  Reference lines are cited at each
// step.  Plain C++ (fp32), compiled by hipcc for gfx950 and by g++ for the host harness.
#pragma once

#include "epsm_path_core.h"
#include "epsm_cp_core.h"
#include "epsm_scatter_core.h"
#include "epsm_tangent_core.h"
#include "../../include/epsm_trace.h"

namespace epsm {

typedef V3<float> F3;
EPSM_HD F3 f3(float x, float y, float z) { return mk3<float>(x, y, z); }
EPSM_HD F3 ld3(const float *p) { return f3(p[0], p[1], p[2]); }
EPSM_HD F3 mul3(F3 a, F3 b) { return f3(a.x * b.x, a.y * b.y, a.z * b.z); }
EPSM_HD float max3(F3 a) { return fmaxf(a.x, fmaxf(a.y, a.z)); }
EPSM_HD float safe_sqrt(float x) { return sqrtf(fmaxf(x, 0.f)); }
EPSM_HD float sqr(float x) { return x * x; }
EPSM_HD F3 normalize3(F3 v) { return v * (1.f / sqrtf(dot(v, v))); }
EPSM_HD float mulsign(float x, float s) { return s < 0.f ? -x : x; }   // x * sign(s), sign(+-0) by bit in drjit; 0 -> +

constexpr float kPi = 3.14159265358979323846f;
constexpr float kInvPi = 0.31830988618379067154f;
constexpr float kEps = 5.9604644775390625e-8f;          // numeric_limits<float>::epsilon()/2  (math.h Epsilon)
constexpr float kRayEps = kEps * 1500.f;                // math::RayEpsilon
constexpr float kShadowEps = kRayEps * 10.f;            // math::ShadowEpsilon
constexpr float kInf = 3.402823466e+38f;

// BSDF flag words (include/mitsuba/render/bsdf.h:31-101)
constexpr uint32_t kFlDiffuseRefl = 0x2, kFlGlossyRefl = 0x8, kFlDeltaRefl = 0x20, kFlDeltaTrans = 0x40,
                   kFlNonSymmetric = 0x4000, kFlFront = 0x8000, kFlBack = 0x10000;
constexpr uint32_t kFlSmooth = 0x2 | 0x4 | 0x8 | 0x10;  // Diffuse | Glossy
constexpr uint32_t kFlDelta = 0x1 | 0x20 | 0x40;        // Null | DeltaReflection | DeltaTransmission

// ---------------------------------------------------------------------------
// sampler: independent (PCG32) seeded through TEA  (src/render/sampler.cpp:115-134)
// ---------------------------------------------------------------------------
struct Pcg32 {
    uint64_t state, inc;
    EPSM_HD uint32_t next_u32() {
        const uint64_t old = state;
        state = old * 0x5851f42d4c957f2dULL + inc;
        const uint32_t xorshifted = (uint32_t) (((old >> 18u) ^ old) >> 27u);
        const uint32_t rot = (uint32_t) (old >> 59u);
        return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
    }
    EPSM_HD float next_1d() {
        union { uint32_t u; float f; } c;
        c.u = (next_u32() >> 9) | 0x3f800000u;
        return c.f - 1.f;
    }
};
EPSM_HD void sample_tea_32(uint32_t v0, uint32_t v1, uint32_t &o0, uint32_t &o1) {
    uint32_t sum = 0;
    for (int i = 0; i < 4; ++i) {
        sum += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + sum) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + sum) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    o0 = v0; o1 = v1;
}
// PCG32::seed(size, initstate, initseq) of Dr.Jit (the published pcg32_srandom_r)
EPSM_HD Pcg32 pcg32_seed(uint64_t initstate, uint64_t initseq) {
    Pcg32 r;
    r.state = 0; r.inc = (initseq << 1) | 1u;
    r.next_u32();
    r.state += initstate;
    r.next_u32();
    return r;
}
EPSM_HD Pcg32 seed_sampler(uint32_t seed, uint32_t wavefront_index) {
    uint32_t v0, v1;
    sample_tea_32(seed, wavefront_index, v0, v1);       // sampler.cpp:127
    return pcg32_seed(v0, v1);                           // m_rng.seed(1, v0, v1)
}

// ---------------------------------------------------------------------------
// warps (include/mitsuba/core/warp.h)
// ---------------------------------------------------------------------------
EPSM_HD void square_to_uniform_disk_concentric(float u, float v, float &ox, float &oy) {
    const float x = 2.f * u - 1.f, y = 2.f * v - 1.f;
    const bool is_zero = x == 0.f && y == 0.f, q13 = fabsf(x) < fabsf(y);
    const float r = q13 ? y : x, rp = q13 ? x : y;
    float phi = 0.25f * kPi * rp / r;
    if (q13) phi = 0.5f * kPi - phi;
    if (is_zero) phi = 0.f;
    ox = r * cosf(phi); oy = r * sinf(phi);
}
EPSM_HD F3 square_to_cosine_hemisphere(float u, float v) {
    float x, y;
    square_to_uniform_disk_concentric(u, v, x, y);
    return f3(x, y, safe_sqrt(1.f - x * x - y * y));
}
EPSM_HD void square_to_uniform_triangle(float u, float v, float &bx, float &by) {
    const float t = safe_sqrt(1.f - u);
    bx = 1.f - t; by = t * v;
}
// coordinate_system (include/mitsuba/core/vector.h, Duff et al.)
EPSM_HD void coordinate_system(F3 n, F3 &s, F3 &t) {
    const float sign = n.z >= 0.f ? 1.f : -1.f, a = -1.f / (sign + n.z), b = n.x * n.y * a;
    s = f3(mulsign(n.x * n.x * a, n.z) + 1.f, mulsign(b, n.z), mulsign(-n.x, n.z));
    t = f3(b, n.y * n.y * a + sign, -n.y);
}

// ---------------------------------------------------------------------------
// ray / triangle / BVH
// ---------------------------------------------------------------------------
struct Ray { F3 o, d; float maxt; };

// Moeller-Trumbore (include/mitsuba/render/mesh.h:343-365)
EPSM_HD bool moeller_trumbore(const Ray &r, F3 p0, F3 p1, F3 p2, float &t, float &u, float &v) {
    const F3 e1 = p1 - p0, e2 = p2 - p0;
    const F3 pvec = cross(r.d, e2);
    const float inv_det = 1.f / dot(e1, pvec);
    const F3 tvec = r.o - p0;
    u = dot(tvec, pvec) * inv_det;
    const F3 qvec = cross(tvec, e1);
    v = dot(r.d, qvec) * inv_det;
    t = dot(e2, qvec) * inv_det;
    return u >= 0.f && u <= 1.f && v >= 0.f && u + v <= 1.f && t >= 0.f && t <= r.maxt;
}
// (slab test of a box: per axis the two plane distances, a min and a max; then max3 / min3 -- in trav_round, four boxes
// per node; `inv_d` is finite (trav_begin), so no 0 * inf turns up there; the far distance is widened by 2 ulp.)
struct TriHit { bool hit; uint32_t tri; float t, u, v; };

// Traversal stack of one path: entry k lives at base[k * stride] (LDS on the GPU: one column per thread,
// conflict-free; a local array on the host).  Depth of the tree <= kBvhStack.
constexpr int kBvhStack = 48;          // a four-wide node pushes up to three references; depth <= 16
constexpr int32_t kBvhNone = 0x7fffffff;
// The first `cap` entries live at base (LDS); a stack that grows beyond them continues at ovf[(k - cap) * ovf_stride]
// (global memory, rarely reached: the wavefront kernels keep only 16 entries per path in LDS to run 8 waves per SIMD).
struct BvhStack {
    uint32_t *base; int stride;
    int cap = kBvhStack; uint32_t *ovf = nullptr; int64_t ovf_stride = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    // `base` is LDS on the device: say so, or the two-way choice below becomes ONE flat load / store whose address is
    // selected (flat accesses to LDS take the long way through the texture path and wait on both counters)
    typedef __attribute__((address_space(3))) uint32_t LdsWord;
    EPSM_HD void put(int k, uint32_t v) const { if (k < cap) ((LdsWord *) base)[k * stride] = v; else ovf[(int64_t) (k - cap) * ovf_stride] = v; }
    EPSM_HD uint32_t get(int k) const { if (k < cap) return ((LdsWord *) base)[k * stride]; return ovf[(int64_t) (k - cap) * ovf_stride]; }
#else
    EPSM_HD void put(int k, uint32_t v) const { if (k < cap) base[k * stride] = v; else ovf[(int64_t) (k - cap) * ovf_stride] = v; }
    EPSM_HD uint32_t get(int k) const { return k < cap ? base[k * stride] : ovf[(int64_t) (k - cap) * ovf_stride]; }
#endif
};

// Ordered traversal of the FOUR-wide BVH (round 3; two-wide before): one 128-byte node holds the boxes of up to four
// children, one component of all four per 16-byte quad; the hit children are sorted by entry distance (five
// compare-exchanges), the nearest is followed and the others pushed farthest first.  Half the dependent node loads of the
// two-wide tree for the same box arithmetic per level pair -- the traversal of incoherent rays waits for exactly those loads.
// While-while: a ROUND descends through inner nodes until the current
// reference is a leaf, then intersects that leaf's triangles -- the lanes of a wave run the (long) triangle code
// together instead of each one in the middle of its own descent.  A reference is a node index (>= 0), a leaf
// ~((first << 3) | count) (< 0) or kBvhNone.  The traversal is a resumable object so that the wavefront kernels can
// give a lane whose ray is finished a new ray between two rounds.
struct Traversal {
    Ray r;
    F3 inv_d, noid;               // 1 / d and -o / d: a slab plane's distance is one fma, (plane) * inv_d + noid
    int32_t cur, best_e;
    int sp;
    TriHit best;
};
EPSM_HD void trav_begin(Traversal &T, const EpsmScene &S, const Ray &r) {
    T.r = r;
    T.best.hit = false; T.best.tri = 0; T.best.t = r.maxt; T.best.u = T.best.v = 0.f;
    // finite reciprocals: a ray parallel to a slab sees its planes at +-1e18 x distance (or at 0 when it lies IN one,
    // which counts as inside) instead of +-inf / NaN
    T.inv_d = f3(fminf(fmaxf(1.f / r.d.x, -1e18f), 1e18f), fminf(fmaxf(1.f / r.d.y, -1e18f), 1e18f), fminf(fmaxf(1.f / r.d.z, -1e18f), 1e18f));
    T.noid = f3(-r.o.x * T.inv_d.x, -r.o.y * T.inv_d.y, -r.o.z * T.inv_d.z);
    T.best_e = -1;
    T.cur = S.n_nodes > 0 ? 0 : kBvhNone;
    T.sp = 0;
}
EPSM_HD bool trav_done(const Traversal &T) { return T.cur == kBvhNone; }
#if defined(EPSM_TRAV_STATS) && !defined(__HIP_DEVICE_COMPILE__)
// host harness only (tools/count_traversal.py): nodes fetched / triangles tested / rays, by ray kind (0 closest hit, 1 any hit)
extern "C" { extern long long g_trav_nodes[2], g_trav_tris[2], g_trav_rays[2]; }
#define EPSM_STAT(x) x
#else
#define EPSM_STAT(x)
#endif
template <bool ANY_HIT>
EPSM_HD void trav_round(Traversal &T, const EpsmScene &S, const BvhStack &st) {
    while (T.cur >= 0 && T.cur != kBvhNone) {
        const EpsmBvhNode n = S.bvh[T.cur];
        EPSM_STAT(__atomic_fetch_add(&g_trav_nodes[ANY_HIT], 1, __ATOMIC_RELAXED);)
        // slab test of the four boxes (an absent child's box is empty and its reference kBvhNone)
        float t[4]; int32_t c[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float ax = fmaf(n.lox[s], T.inv_d.x, T.noid.x), bx = fmaf(n.hix[s], T.inv_d.x, T.noid.x);
            const float ay = fmaf(n.loy[s], T.inv_d.y, T.noid.y), by = fmaf(n.hiy[s], T.inv_d.y, T.noid.y);
            const float az = fmaf(n.loz[s], T.inv_d.z, T.noid.z), bz = fmaf(n.hiz[s], T.inv_d.z, T.noid.z);
            const float t0 = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.f));
            const float t1 = fminf(fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz)) * 1.0000004f, T.r.maxt);
            const bool h = (t0 <= t1) & (n.c[s] != kBvhNone);
            t[s] = h ? t0 : kInf;
            c[s] = h ? n.c[s] : kBvhNone;
        }
        // sort by entry distance (selects, no branches: the lanes of a wave take every outcome anyway)
#define EPSM_CX(i, j) { const bool sw = t[j] < t[i]; const float ta = sw ? t[j] : t[i], tb = sw ? t[i] : t[j]; \
                        const int32_t ca = sw ? c[j] : c[i], cb = sw ? c[i] : c[j]; t[i] = ta; t[j] = tb; c[i] = ca; c[j] = cb; }
        EPSM_CX(0, 1) EPSM_CX(2, 3) EPSM_CX(0, 2) EPSM_CX(1, 3) EPSM_CX(1, 2)
#undef EPSM_CX
        if (c[3] != kBvhNone && T.sp < kBvhStack) st.put(T.sp++, (uint32_t) c[3]);
        if (c[2] != kBvhNone && T.sp < kBvhStack) st.put(T.sp++, (uint32_t) c[2]);
        if (c[1] != kBvhNone && T.sp < kBvhStack) st.put(T.sp++, (uint32_t) c[1]);
        if (c[0] != kBvhNone) T.cur = c[0];
        else T.cur = T.sp > 0 ? (int32_t) st.get(--T.sp) : kBvhNone;
    }
    if (T.cur == kBvhNone) return;
    const uint32_t ref = ~(uint32_t) T.cur;
    const int32_t first = (int32_t) (ref >> 3), count = (int32_t) (ref & 7u);
    for (int32_t e = first; e < first + count; ++e) {
        const float *q = S.tri_verts + 9 * (int64_t) e;
        float t, u, v;
        EPSM_STAT(__atomic_fetch_add(&g_trav_tris[ANY_HIT], 1, __ATOMIC_RELAXED);)
        if (moeller_trumbore(T.r, ld3(q), ld3(q + 3), ld3(q + 6), t, u, v)) {
            T.best.hit = true; T.best_e = e; T.best.t = t; T.best.u = u; T.best.v = v;
            T.r.maxt = t;
            if (ANY_HIT) break;
        }
    }
    if (ANY_HIT && T.best.hit) { T.cur = kBvhNone; return; }
    T.cur = T.sp > 0 ? (int32_t) st.get(--T.sp) : kBvhNone;
}
EPSM_HD TriHit trav_result(const Traversal &T, const EpsmScene &S) {
    TriHit best = T.best;
    if (best.hit) best.tri = S.prim_index[T.best_e];
    return best;
}
template <bool ANY_HIT>
EPSM_HD TriHit intersect(const EpsmScene &S, Ray r, const BvhStack &st) {
    Traversal T;
    trav_begin(T, S, r);
    EPSM_STAT(__atomic_fetch_add(&g_trav_rays[ANY_HIT], 1, __ATOMIC_RELAXED);)
    while (!trav_done(T)) trav_round<ANY_HIT>(T, S, st);
    return trav_result(T, S);
}

// ---------------------------------------------------------------------------
// surface interaction (src/render/mesh.cpp:632-892 incl. the EPSM fields :712-720, 784-827)
// ---------------------------------------------------------------------------
struct SurfHit {
    bool valid;
    float t;
    uint32_t tri, mesh, vi[3], mesh_flags;
    int32_t bsdf, emitter;
    float b0, b1, b2;                 // weights of p0,p1,p2: b1 = u, b2 = v, b0 = 1-u-v (mesh.cpp:698-700)
    F3 p, n, shn, fs, ft;             // position, geometric normal, shading frame (s, t, n)
    F3 p0, p1, p2, n0, n1, n2;        // logged triangle (normals post-flip, flat: n0=n1=n2=n)
    F3 wi;                            // -ray.d in the shading frame
    float uvx, uvy;                   // si.uv: interpolated texture coordinates, (b1, b2) for a mesh without any (mesh.cpp:736-745)
};
EPSM_HD F3 to_local(const SurfHit &h, F3 v) { return f3(dot(v, h.fs), dot(v, h.ft), dot(v, h.shn)); }
EPSM_HD F3 to_world(const SurfHit &h, F3 v) { return h.fs * v.x + h.ft * v.y + h.shn * v.z; }

EPSM_HD SurfHit surface_interaction(const EpsmScene &S, const Ray &r, const TriHit &th) {
    SurfHit h;
    h.valid = th.hit;
    h.t = th.hit ? th.t : kInf;
    h.tri = th.tri; h.mesh = 0; h.mesh_flags = 0; h.bsdf = -1; h.emitter = -1;
    h.vi[0] = h.vi[1] = h.vi[2] = kNoIndex;
    h.b0 = h.b1 = h.b2 = 0.f;
    h.p = h.n = h.shn = h.fs = h.ft = h.wi = zero3<float>();
    h.p0 = h.p1 = h.p2 = h.n0 = h.n1 = h.n2 = zero3<float>();
    h.uvx = h.uvy = 0.f;
    if (!th.hit) return h;
    const uint32_t *iv = S.tri + 3 * (int64_t) th.tri;
    h.vi[0] = iv[0]; h.vi[1] = iv[1]; h.vi[2] = iv[2];
    h.mesh = S.tri_mesh[th.tri];
    const EpsmMesh m = S.meshes[h.mesh];
    h.mesh_flags = m.flags; h.bsdf = m.bsdf; h.emitter = m.emitter;
    h.p0 = ld3(S.positions + 3 * (int64_t) iv[0]);
    h.p1 = ld3(S.positions + 3 * (int64_t) iv[1]);
    h.p2 = ld3(S.positions + 3 * (int64_t) iv[2]);
    h.b1 = th.u; h.b2 = th.v; h.b0 = 1.f - th.u - th.v;
    h.uvx = h.b1; h.uvy = h.b2;
    if ((m.flags & EPSM_MESH_HAS_UV) && S.texcoords) {
        const float *t0 = S.texcoords + 2 * (int64_t) iv[0], *t1 = S.texcoords + 2 * (int64_t) iv[1], *t2 = S.texcoords + 2 * (int64_t) iv[2];
        h.uvx = t0[0] * h.b0 + t1[0] * h.b1 + t2[0] * h.b2; h.uvy = t0[1] * h.b0 + t1[1] * h.b1 + t2[1] * h.b2;
    }
    h.p = h.p0 * h.b0 + h.p1 * h.b1 + h.p2 * h.b2;                       // mesh.cpp:709
    h.n = normalize3(cross(h.p1 - h.p0, h.p2 - h.p0));                   // mesh.cpp:729
    if (m.flags & EPSM_MESH_VERTEX_NORMALS) {                           // mesh.cpp:784-790
        h.n0 = ld3(S.normals + 3 * (int64_t) iv[0]);
        h.n1 = ld3(S.normals + 3 * (int64_t) iv[1]);
        h.n2 = ld3(S.normals + 3 * (int64_t) iv[2]);
        h.shn = normalize3(h.n0 * h.b0 + h.n1 * h.b1 + h.n2 * h.b2);
    } else {                                                            // mesh.cpp:811-816
        h.shn = h.n; h.n0 = h.n1 = h.n2 = h.n;
    }
    if (m.flags & EPSM_MESH_FLIP_NORMALS) {                             // mesh.cpp:820-827
        h.n = -h.n; h.shn = -h.shn; h.n0 = -h.n0; h.n1 = -h.n1; h.n2 = -h.n2;
    }
    // SurfaceInteraction::initialize_sh_frame with dp_du from coordinate_system(n) (mesh.cpp:734)
    F3 dpdu, dpdv;
    coordinate_system(h.n, dpdu, dpdv);
    h.fs = normalize3(dpdu - h.shn * dot(h.shn, dpdu));
    h.ft = cross(h.shn, h.fs);
    h.wi = to_local(h, -r.d);
    return h;
}
// Interaction::offset_p / spawn_ray / spawn_ray_to (include/mitsuba/render/interaction.h)
EPSM_HD F3 offset_p(const SurfHit &h, F3 d) {
    float mag = (1.f + fmaxf(fabsf(h.p.x), fmaxf(fabsf(h.p.y), fabsf(h.p.z)))) * kRayEps;
    mag = mulsign(mag, dot(h.n, d));
    return h.p + h.n * mag;
}
EPSM_HD Ray spawn_ray(const SurfHit &h, F3 d) { Ray r; r.o = offset_p(h, d); r.d = d; r.maxt = kInf; return r; }
EPSM_HD Ray spawn_ray_to(const SurfHit &h, F3 target, float &dist) {
    Ray r;
    r.o = offset_p(h, target - h.p);
    F3 d = target - r.o;
    dist = sqrtf(dot(d, d));
    r.d = d * (1.f / dist);
    r.maxt = dist * (1.f - kShadowEps);
    return r;
}

// ---------------------------------------------------------------------------
// Fresnel (include/mitsuba/render/fresnel.h:34-72, 92-117)
// ---------------------------------------------------------------------------
EPSM_HD void fresnel(float cos_i, float eta, float &r, float &cos_t, float &eta_it, float &eta_ti) {
    const bool outside = cos_i >= 0.f;
    const float rcp_eta = 1.f / eta;
    eta_it = outside ? eta : rcp_eta;
    eta_ti = outside ? rcp_eta : eta;
    const float cos_t_sqr = 1.f - (1.f - cos_i * cos_i) * eta_ti * eta_ti;
    const float ci = fabsf(cos_i), ct = safe_sqrt(cos_t_sqr);
    const bool index_matched = eta == 1.f, special = index_matched || ci == 0.f;
    const float a_s = (ci - eta_it * ct) / (ci + eta_it * ct);
    const float a_p = (ct - eta_it * ci) / (ct + eta_it * ci);
    r = 0.5f * (a_s * a_s + a_p * a_p);
    if (special) r = index_matched ? 0.f : 1.f;
    cos_t = cos_i >= 0.f ? -ct : ct;                                     // mulsign_neg
}
EPSM_HD float fresnel_conductor(float cos_i, float eta_r, float eta_i) {
    const float c2 = cos_i * cos_i, s2 = 1.f - c2, s4 = s2 * s2;
    const float temp_1 = eta_r * eta_r - eta_i * eta_i - s2;
    const float a_2_pb_2 = safe_sqrt(temp_1 * temp_1 + 4.f * eta_i * eta_i * eta_r * eta_r);
    const float a = safe_sqrt(0.5f * (a_2_pb_2 + temp_1));
    const float term_1 = a_2_pb_2 + c2, term_2 = 2.f * cos_i * a;
    const float r_s = (term_1 - term_2) / (term_1 + term_2);
    const float term_3 = a_2_pb_2 * c2 + s4, term_4 = term_2 * s2;
    const float r_p = r_s * (term_3 - term_4) / (term_3 + term_4);
    return 0.5f * (r_s + r_p);
}
EPSM_HD F3 fresnel_conductor3(float cos_i, const EpsmBsdf &b) {
    return f3(fresnel_conductor(cos_i, b.eta[0], b.k[0]), fresnel_conductor(cos_i, b.eta[1], b.k[1]),
              fresnel_conductor(cos_i, b.eta[2], b.k[2]));
}

// ---------------------------------------------------------------------------
// microfacet distribution (include/mitsuba/render/microfacet.h, isotropic)
// ---------------------------------------------------------------------------
EPSM_HD float mf_eval(const EpsmBsdf &b, F3 m) {
    const float a = b.alpha, ct = m.z, ct2 = ct * ct;
    float result;
    if (b.distr == EPSM_DISTR_BECKMANN)
        result = expf(-(sqr(m.x / a) + sqr(m.y / a)) / ct2) / (kPi * a * a * sqr(ct2));
    else
        result = 1.f / (kPi * a * a * sqr(sqr(m.x / a) + sqr(m.y / a) + sqr(m.z)));
    return result * ct > 1e-20f ? result : 0.f;
}
EPSM_HD float mf_smith_g1(const EpsmBsdf &b, F3 v, F3 m) {
    const float xy_alpha_2 = sqr(b.alpha * v.x) + sqr(b.alpha * v.y), tan2 = xy_alpha_2 / sqr(v.z);
    float result;
    if (b.distr == EPSM_DISTR_BECKMANN) {
        const float a = 1.f / sqrtf(tan2), a2 = a * a;
        result = a >= 1.6f ? 1.f : (3.535f * a + 2.181f * a2) / (1.f + 2.276f * a + 2.577f * a2);
    } else {
        result = 2.f / (1.f + sqrtf(1.f + tan2));
    }
    if (xy_alpha_2 == 0.f) result = 1.f;
    if (dot(v, m) * v.z <= 0.f) result = 0.f;
    return result;
}
EPSM_HD float mf_pdf(const EpsmBsdf &b, F3 wi, F3 m) {
    float result = mf_eval(b, m);
    if (b.sample_visible) result *= mf_smith_g1(b, wi, m) * fabsf(dot(wi, m)) / wi.z;
    else result *= m.z;
    return result;
}
// The fork forces the D*cos sampling branch (`if (true)`, microfacet.h) whatever sample_visible says.
// Also returns d m / d alpha (tan(theta_m) is proportional to alpha for both distributions).
EPSM_HD F3 mf_sample(const EpsmBsdf &b, float u, float v, float &pdf, F3 &dm_dalpha) {
    const float phi = 2.f * kPi * v, sin_phi = sinf(phi), cos_phi = cosf(phi);
    const float alpha_2 = b.alpha * b.alpha;
    float cos_theta, cos_theta_2;
    if (b.distr == EPSM_DISTR_BECKMANN) {
        cos_theta = 1.f / sqrtf(1.f - alpha_2 * logf(1.f - u));
        cos_theta_2 = cos_theta * cos_theta;
        const float c3 = fmaxf(cos_theta_2 * cos_theta, 1e-20f);
        pdf = (1.f - u) / (kPi * alpha_2 * c3);
    } else {
        const float tan2 = alpha_2 * u / (1.f - u);
        cos_theta = 1.f / sqrtf(1.f + tan2);
        cos_theta_2 = cos_theta * cos_theta;
        const float temp = 1.f + tan2 / alpha_2, c3 = fmaxf(cos_theta_2 * cos_theta, 1e-20f);
        pdf = 1.f / (kPi * alpha_2 * c3 * temp * temp);
    }
    const float sin_theta = sqrtf(1.f - cos_theta_2);
    const float dtheta = sin_theta * cos_theta / b.alpha;
    dm_dalpha = f3(cos_phi * cos_theta, sin_phi * cos_theta, -sin_theta) * dtheta;
    return f3(cos_phi * sin_theta, sin_phi * sin_theta, cos_theta);
}

// ---------------------------------------------------------------------------
// BSDFs in the local frame
// ---------------------------------------------------------------------------
struct BsdfSample { F3 wo, weight, hf, dhf; float pdf, eta; uint32_t sampled_type; bool valid; };

EPSM_HD uint32_t bsdf_flags(const EpsmBsdf &b) {
    uint32_t f;
    switch (b.type) {
        case EPSM_BSDF_DIFFUSE_T: f = kFlDiffuseRefl | kFlFront; break;                       // diffuse.cpp
        case EPSM_BSDF_CONDUCTOR_T: f = kFlDeltaRefl | kFlFront; break;                       // conductor.cpp
        case EPSM_BSDF_ROUGHCONDUCTOR_T: f = kFlGlossyRefl | kFlFront; break;                 // roughconductor.cpp
        default: f = kFlDeltaRefl | kFlDeltaTrans | kFlFront | kFlBack | kFlNonSymmetric; break;   // dielectric.cpp
    }
    if (b.twosided) f |= kFlBack;                                                             // twosided.cpp
    return f;
}
EPSM_HD F3 reflect_local(F3 wi) { return f3(-wi.x, -wi.y, wi.z); }
EPSM_HD F3 reflect_about(F3 wi, F3 m) { return m * (2.f * dot(wi, m)) - wi; }

EPSM_HD BsdfSample bsdf_sample(const EpsmBsdf &b, F3 wi_in, float s1, float s2x, float s2y, bool active) {
    BsdfSample o;
    o.wo = o.weight = o.hf = o.dhf = zero3<float>(); o.pdf = 0.f; o.eta = 0.f; o.sampled_type = 0; o.valid = false;
    if (!active) return o;
    F3 wi = wi_in;
    const bool flipped = b.twosided && wi.z < 0.f;                         // twosided.cpp: evaluate the front BSDF mirrored
    if (flipped) wi.z = -wi.z;
    const float cti = wi.z;
    const F3 R = ld3(b.reflectance);
    switch (b.type) {
        case EPSM_BSDF_DIFFUSE_T: {                                       // diffuse.cpp:126-150
            if (cti <= 0.f) return o;
            o.wo = square_to_cosine_hemisphere(s2x, s2y);
            o.pdf = o.wo.z * kInvPi;
            o.eta = 1.f; o.sampled_type = kFlDiffuseRefl;
            o.weight = R;
            o.valid = o.pdf > 0.f;
        } break;
        case EPSM_BSDF_CONDUCTOR_T: {                                     // conductor.cpp:235-270
            if (cti <= 0.f) return o;
            o.wo = reflect_local(wi); o.pdf = 1.f; o.eta = 1.f; o.sampled_type = kFlDeltaRefl;
            o.weight = mul3(fresnel_conductor3(cti, b), R);
            o.valid = true;
        } break;
        case EPSM_BSDF_ROUGHCONDUCTOR_T: {                                // roughconductor.cpp:225-300
            if (cti <= 0.f) return o;
            float pdf; F3 dm;
            const F3 m = mf_sample(b, s2x, s2y, pdf, dm);
            o.wo = reflect_about(wi, m);
            o.eta = 1.f; o.sampled_type = kFlGlossyRefl;
            o.hf = m; o.dhf = dm;                                         // bs.hf = m  (:255)
            if (!(pdf != 0.f && o.wo.z > 0.f)) { o.hf = m; return o; }
            float w;
            if (b.sample_visible) w = mf_smith_g1(b, o.wo, m);
            else w = mf_smith_g1(b, wi, m) * mf_smith_g1(b, o.wo, m) * dot(wi, m) / (cti * m.z);
            o.pdf = pdf / (4.f * dot(o.wo, m));
            o.weight = mul3(fresnel_conductor3(dot(wi, m), b), R) * w;
            o.valid = true;
        } break;
        default: {                                                        // dielectric.cpp:250-330
            const float eta = b.int_ior / b.ext_ior;
            float r_i, cos_t, eta_it, eta_ti;
            fresnel(wi_in.z, eta, r_i, cos_t, eta_it, eta_ti);
            const bool selected_r = s1 <= r_i;
            o.pdf = selected_r ? r_i : 1.f - r_i;
            o.sampled_type = selected_r ? kFlDeltaRefl : kFlDeltaTrans;
            o.wo = selected_r ? reflect_local(wi_in) : f3(-eta_ti * wi_in.x, -eta_ti * wi_in.y, cos_t);
            o.eta = selected_r ? 1.f : eta_it;
            o.weight = selected_r ? R : f3(1.f, 1.f, 1.f) * (eta_ti * eta_ti);   // radiance transport: * sqr(eta_ti)
            o.valid = o.pdf > 0.f;
            return o;                                                     // never two-sided
        }
    }
    if (flipped) { o.wo.z = -o.wo.z; }
    return o;
}
// value (incl. cosine) and pdf for a given direction (eval_pdf)
EPSM_HD void bsdf_eval_pdf(const EpsmBsdf &b, F3 wi, F3 wo, F3 &value, float &pdf) {
    value = zero3<float>(); pdf = 0.f;
    if (b.twosided && wi.z < 0.f) { wi.z = -wi.z; wo.z = -wo.z; }
    const float cti = wi.z, cto = wo.z;
    if (!(cti > 0.f && cto > 0.f)) return;
    if (b.type == EPSM_BSDF_DIFFUSE_T) {                                  // diffuse.cpp:152-190
        value = ld3(b.reflectance) * (kInvPi * cto);
        pdf = cto * kInvPi;
    } else if (b.type == EPSM_BSDF_ROUGHCONDUCTOR_T) {                    // roughconductor.cpp:302-400
        const F3 H = normalize3(wi + wo);
        const float D = mf_eval(b, H);
        if (D == 0.f) return;
        const float G = mf_smith_g1(b, wi, H) * mf_smith_g1(b, wo, H);
        const float result = D * G / (4.f * cti);
        value = mul3(fresnel_conductor3(dot(wi, H), b), ld3(b.reflectance)) * result;
        pdf = mf_pdf(b, wi, H) / (4.f * dot(wo, H));
    }
}

// ---------------------------------------------------------------------------
// `bitmap` textures (src/textures/bitmap.cpp:366-418, 585-625): value at (u, v) and -- for the reparameterised pass, where the
// point a ray sees slides over the texture -- its derivative w.r.t. (u, v)
// ---------------------------------------------------------------------------
EPSM_HD int tex_wrap(int i, int n) { int r = i % n; return r < 0 ? r + n : r; }
EPSM_HD F3 tex_eval(const EpsmTexture &T, float u, float v, F3 *du = nullptr, F3 *dv = nullptr) {
    const float x = u * (float) T.width - 0.5f, y = v * (float) T.height - 0.5f;
    if (du) *du = zero3<float>();
    if (dv) *dv = zero3<float>();
    if (T.nearest) {
        const int i = tex_wrap((int) floorf(x + 0.5f), T.width), j = tex_wrap((int) floorf(y + 0.5f), T.height);
        return ld3(T.texels + 3 * ((int64_t) j * T.width + i));
    }
    const float fxf = floorf(x), fyf = floorf(y);
    const int i0 = tex_wrap((int) fxf, T.width), j0 = tex_wrap((int) fyf, T.height), i1 = tex_wrap((int) fxf + 1, T.width),
              j1 = tex_wrap((int) fyf + 1, T.height);
    const float fx = x - fxf, fy = y - fyf;
    const F3 a = ld3(T.texels + 3 * ((int64_t) j0 * T.width + i0)), b = ld3(T.texels + 3 * ((int64_t) j0 * T.width + i1)),
             c = ld3(T.texels + 3 * ((int64_t) j1 * T.width + i0)), e = ld3(T.texels + 3 * ((int64_t) j1 * T.width + i1));
    if (du) *du = ((b - a) * (1.f - fy) + (e - c) * fy) * (float) T.width;
    if (dv) *dv = ((c - a) * (1.f - fx) + (e - b) * fx) * (float) T.height;
    return a * ((1.f - fx) * (1.f - fy)) + b * (fx * (1.f - fy)) + c * ((1.f - fx) * fy) + e * (fx * fy);
}

// ---------------------------------------------------------------------------
// emitters (src/emitters/area.cpp, point.cpp; src/render/scene.cpp:226-300; src/render/mesh.cpp sample_position)
// ---------------------------------------------------------------------------
struct EmitterSample {
    F3 p, n, d, weight;              // ds.p, ds.n, ds.d, radiance / pdf (already zero when occluded / facing away)
    float pdf, dist;
    bool delta, valid;
    uint32_t tri; float b0, b1;      // triangle id + barycentrics of the sampled point (parameter addressing)
    int emitter;                     // index of the sampled emitter
};
EPSM_HD F3 emitter_normal(const EpsmScene &S, const EpsmMesh &m, const uint32_t *iv, float w0, float w1, float w2, F3 q0, F3 q1, F3 q2) {
    F3 n = normalize3(cross(q1 - q0, q2 - q0));
    if (m.flags & EPSM_MESH_VERTEX_NORMALS)
        n = normalize3(ld3(S.normals + 3 * (int64_t) iv[0]) * w0 + ld3(S.normals + 3 * (int64_t) iv[1]) * w1 +
                       ld3(S.normals + 3 * (int64_t) iv[2]) * w2);
    if (m.flags & EPSM_MESH_FLIP_NORMALS) n = -n;
    return n;
}
// ---- the environment emitter (include/epsm_trace.h, EpsmEnvironment; src/emitters/constant.cpp, envmap.cpp)
EPSM_HD bool has_environment(const EpsmScene &S) { return S.env.kind != EPSM_ENV_NONE; }
EPSM_HD F3 env_to_local(const EpsmEnvironment &E, F3 d) {
    return f3(E.to_local[0] * d.x + E.to_local[1] * d.y + E.to_local[2] * d.z, E.to_local[3] * d.x + E.to_local[4] * d.y + E.to_local[5] * d.z,
              E.to_local[6] * d.x + E.to_local[7] * d.y + E.to_local[8] * d.z);
}
EPSM_HD F3 env_to_world(const EpsmEnvironment &E, F3 d) {                  // the transpose: a rotation
    return f3(E.to_local[0] * d.x + E.to_local[3] * d.y + E.to_local[6] * d.z, E.to_local[1] * d.x + E.to_local[4] * d.y + E.to_local[7] * d.z,
              E.to_local[2] * d.x + E.to_local[5] * d.y + E.to_local[8] * d.z);
}
// continuous cell coordinates of a direction of the emitter's frame: x in [0, width), y in [0, height - 1]  (envmap.cpp:416-422)
EPSM_HD void env_cell_coords(const EpsmEnvironment &E, F3 v, float &x, float &y) {
    float u = atan2f(v.x, -v.z) * (0.5f / kPi) - 0.5f / (float) E.width;
    u -= floorf(u);
    const float w = acosf(fminf(fmaxf(v.y, -1.f), 1.f)) * (1.f / kPi);
    x = fminf(u * (float) E.width, (float) E.width - 1e-3f);
    y = fminf(w * (float) (E.height - 1), (float) (E.height - 1));
}
// radiance seen along the WORLD direction d (a ray that left the scene)
EPSM_HD F3 env_eval(const EpsmScene &S, F3 d) {
    const EpsmEnvironment &E = S.env;
    if (E.kind == EPSM_ENV_CONSTANT) return ld3(S.emitters[E.emitter].radiance);
    float x, y;
    env_cell_coords(E, env_to_local(E, d), x, y);
    const int i = (int) x, j = (int) fminf(y, (float) (E.height - 2));
    const float fx = x - (float) i, fy = y - (float) j;
    const int64_t row = (int64_t) (E.width + 1) * 3;
    const float *t00 = E.texels + j * row + 3 * (int64_t) i, *t01 = t00 + 3, *t10 = t00 + row, *t11 = t10 + 3;
    const float w00 = (1.f - fx) * (1.f - fy), w01 = fx * (1.f - fy), w10 = (1.f - fx) * fy, w11 = fx * fy;
    return ld3(t00) * w00 + ld3(t01) * w01 + ld3(t10) * w10 + ld3(t11) * w11;
}
// the same with its derivative w.r.t. the world direction (the map is bilinear in (phi, theta): envmap.cpp evaluates it through a
// differentiable texture lookup, which the reparameterised pass needs where its warp field turns a direction); g[c] = d L_c / d d
EPSM_HD F3 env_eval_grad(const EpsmScene &S, F3 d, F3 g[3]) {
    const EpsmEnvironment &E = S.env;
    g[0] = g[1] = g[2] = zero3<float>();
    if (E.kind == EPSM_ENV_CONSTANT) return ld3(S.emitters[E.emitter].radiance);
    const F3 v = env_to_local(E, d);
    float x, y;
    env_cell_coords(E, v, x, y);
    const int i = (int) x, j = (int) fminf(y, (float) (E.height - 2));
    const float fx = x - (float) i, fy = y - (float) j;
    const int64_t row = (int64_t) (E.width + 1) * 3;
    const float *t00 = E.texels + j * row + 3 * (int64_t) i, *t01 = t00 + 3, *t10 = t00 + row, *t11 = t10 + 3;
    const F3 a = ld3(t00), b = ld3(t01), c = ld3(t10), e = ld3(t11);
    const F3 L = a * ((1.f - fx) * (1.f - fy)) + b * (fx * (1.f - fy)) + c * ((1.f - fx) * fy) + e * (fx * fy);
    const F3 Lx = (b - a) * (1.f - fy) + (e - c) * fy, Ly = (c - a) * (1.f - fx) + (e - b) * fx;      // per cell coordinate
    // x = W (atan2(v.x, -v.z) / 2 pi - ...), y = (H - 1) acos(v.y) / pi
    const float r2 = fmaxf(v.x * v.x + v.z * v.z, 1e-12f);
    const float xs = (float) E.width * (0.5f / kPi) / r2, ys = -(float) (E.height - 1) * (1.f / kPi) / sqrtf(r2);
    const float dx_vx = -v.z * xs, dx_vz = v.x * xs, dy_vy = ys;
    const float Lxc[3] = {Lx.x, Lx.y, Lx.z}, Lyc[3] = {Ly.x, Ly.y, Ly.z};
    for (int ch = 0; ch < 3; ++ch) g[ch] = env_to_world(E, f3(Lxc[ch] * dx_vx, Lyc[ch] * dy_vy, Lxc[ch] * dx_vz));
    return L;
}
// density, per solid angle, of sample_environment producing the WORLD direction d (the emitter choice not included)
EPSM_HD float env_pdf(const EpsmScene &S, F3 d) {
    const EpsmEnvironment &E = S.env;
    if (E.kind == EPSM_ENV_CONSTANT) return 0.25f / kPi;
    const F3 v = env_to_local(E, d);
    float x, y;
    env_cell_coords(E, v, x, y);
    const int i = (int) x, j = (int) fminf(y, (float) (E.height - 2));
    const float inv_sin = 1.f / sqrtf(fmaxf(v.x * v.x + v.z * v.z, 1e-14f));                 // envmap.cpp:401-402
    return E.cell_pdf[(int64_t) j * E.width + i] * inv_sin * (1.f / (2.f * kPi * kPi));
}
// direction + density of an environment sample (u, v uniform in [0,1)); false: the map is black
EPSM_HD bool env_sample(const EpsmScene &S, float u, float v, F3 &d, float &pdf) {
    const EpsmEnvironment &E = S.env;
    if (E.kind == EPSM_ENV_CONSTANT) {                                       // warp.h square_to_uniform_sphere
        const float z = 1.f - 2.f * v, r = safe_sqrt(1.f - z * z), phi = 2.f * kPi * u;
        d = f3(r * cosf(phi), r * sinf(phi), z);
        pdf = 0.25f / kPi;
        return true;
    }
    const int rows = E.height - 1;
    int lo = 0, hi = rows - 1;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (E.row_cdf[mid] < v) lo = mid + 1; else hi = mid; }
    const int j = lo;
    const float r0 = j > 0 ? E.row_cdf[j - 1] : 0.f, r1 = E.row_cdf[j];
    const float fy = r1 > r0 ? fminf((v - r0) / (r1 - r0), 0.999999f) : 0.5f;
    const float *cc = E.col_cdf + (int64_t) j * E.width;
    lo = 0; hi = E.width - 1;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (cc[mid] < u) lo = mid + 1; else hi = mid; }
    const int i = lo;
    const float c0 = i > 0 ? cc[i - 1] : 0.f, c1 = cc[i];
    const float fx = c1 > c0 ? fminf((u - c0) / (c1 - c0), 0.999999f) : 0.5f;
    const float cell = E.cell_pdf[(int64_t) j * E.width + i];
    if (!(cell > 0.f)) { pdf = 0.f; d = f3(0.f, 1.f, 0.f); return false; }
    const float uu = ((float) i + fx + 0.5f) / (float) E.width, vv = ((float) j + fy) / (float) rows;
    const float theta = vv * kPi, phi = uu * 2.f * kPi;
    const float st = sinf(theta);
    const F3 l = f3(sinf(phi) * st, cosf(theta), -cosf(phi) * st);           // envmap.cpp:392-395
    pdf = cell / (2.f * kPi * kPi * fmaxf(st, 1e-7f));
    d = env_to_world(E, l);
    return true;
}

// `test_now` = false leaves the visibility test (scene.cpp:270-275) to the caller, which may know that the sample
// contributes nothing whatever the test says (path_bounce).
template <class Vis>
EPSM_HD EmitterSample sample_emitter_direction(const EpsmScene &S, const SurfHit &ref, float u, float v, bool active,
                                               Vis &vis, bool test_now = true) {
    EmitterSample e;
    e.p = e.n = e.d = e.weight = zero3<float>(); e.pdf = 0.f; e.dist = 0.f; e.delta = false; e.valid = false;
    e.tri = kNoIndex; e.b0 = e.b1 = 0.f; e.emitter = -1;
    if (!active || S.n_emitters <= 0) return e;
    // scene.cpp:233-246: uniform emitter choice, the sample is re-used
    const int count = S.n_emitters;
    int index = 0;
    float emitter_weight = 1.f;
    if (count > 1) {
        index = (int) fminf(u * (float) count, (float) (count - 1));
        u = (u - index / (float) count) * count;
        emitter_weight = (float) count;
    }
    const EpsmEmitter em = S.emitters[index];
    e.emitter = (int) index;
    F3 radiance = ld3(em.radiance);
    if (em.type == EPSM_EMITTER_CONSTANT || em.type == EPSM_EMITTER_ENVMAP) {   // constant.cpp:105-133, envmap.cpp:380-418
        F3 d; float pdf;
        // (an environment-typed emitter that is not THE environment of EpsmScene.env has no tables: it yields no sample)
        if (has_environment(S) && (int) index == S.env.emitter && env_sample(S, u, v, d, pdf)) {
            const F3 off = ref.p - ld3(S.env.center);
            e.dist = 2.f * fmaxf(S.env.radius, sqrtf(dot(off, off)));
            e.d = d; e.p = ref.p + d * e.dist; e.n = -d;
            e.pdf = pdf;
            e.weight = env_eval(S, d) * (1.f / pdf);
        }
    } else if (em.type == EPSM_EMITTER_POINT) {                           // point.cpp:96-115
        e.p = ld3(em.position);
        F3 d = e.p - ref.p;
        const float dist2 = dot(d, d);
        e.dist = sqrtf(dist2); e.d = d * (1.f / e.dist);
        e.pdf = 1.f; e.delta = true; e.n = zero3<float>();
        e.weight = radiance * (1.f / dist2);
    } else {                                                              // area.cpp:117-146 + Mesh::sample_position
        const EpsmMesh m = S.meshes[em.mesh];
        // triangle by area (sample_reuse on sample.y in Mitsuba; here on v), binary search in the CDF
        const float *cdf = S.emitter_cdf + m.cdf_begin;
        uint32_t lo = 0, hi = m.tri_count - 1;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (cdf[mid] < v) lo = mid + 1; else hi = mid; }
        const float c0 = lo > 0 ? cdf[lo - 1] : 0.f, c1 = cdf[lo];
        v = c1 > c0 ? (v - c0) / (c1 - c0) : 0.f;
        const uint32_t t = m.tri_begin + lo;
        const uint32_t *iv = S.tri + 3 * (int64_t) t;
        const F3 q0 = ld3(S.positions + 3 * (int64_t) iv[0]), q1 = ld3(S.positions + 3 * (int64_t) iv[1]),
                 q2 = ld3(S.positions + 3 * (int64_t) iv[2]);
        float bx, by;
        square_to_uniform_triangle(u, v, bx, by);
        const float w0 = 1.f - bx - by;
        e.p = q0 * w0 + q1 * bx + q2 * by;
        e.n = emitter_normal(S, m, iv, w0, bx, by, q0, q1, q2);
        e.tri = t; e.b0 = w0; e.b1 = bx;
        F3 d = e.p - ref.p;
        const float dist2 = dot(d, d);
        e.dist = sqrtf(dist2); e.d = d * (1.f / e.dist);
        const float dp = fabsf(dot(e.d, e.n));
        e.pdf = dp != 0.f ? (1.f / m.area) * dist2 / dp : 0.f;           // shape.cpp sample_direction
        const bool facing = dot(e.d, e.n) < 0.f && e.pdf != 0.f;          // area.cpp:129
        e.weight = facing ? radiance * (1.f / e.pdf) : zero3<float>();
        if (!facing) e.pdf = 0.f;
    }
    e.pdf /= emitter_weight;                                              // scene.cpp:262-266
    e.weight = e.weight * emitter_weight;
    e.valid = e.pdf != 0.f;
    if (e.valid && test_now) {                                            // scene.cpp:270-275 test_visibility
        float dist;
        const Ray sr = spawn_ray_to(ref, e.p, dist);
        if (vis.occluded(S, sr)) e.weight = zero3<float>();
    }
    return e;
}
// pdf of having sampled the point `h` on an emitter from `ref` (scene.cpp pdf_emitter_direction)
EPSM_HD float pdf_emitter_direction(const EpsmScene &S, F3 ref_p, const SurfHit &h) {
    if (h.emitter < 0) return 0.f;
    const EpsmMesh m = S.meshes[h.mesh];
    F3 d = h.p - ref_p;
    const float dist2 = dot(d, d);
    d = d * (1.f / sqrtf(dist2));
    const float dp = fabsf(dot(d, h.shn));                                // DirectionSample(si): n = si.sh_frame.n
    float pdf = dp != 0.f ? (1.f / m.area) * dist2 / dp : 0.f;
    if (S.n_emitters > 1) pdf /= (float) S.n_emitters;
    return pdf;
}
EPSM_HD float mis_weight(float pdf_a, float pdf_b) {                      // common.py:1224-1230
    const float a2 = pdf_a * pdf_a;
    return pdf_a > 0.f ? a2 / (pdf_b * pdf_b + a2) : 0.f;
}

// ---------------------------------------------------------------------------
// camera (src/sensors/perspective.cpp:238-279) and sample_rays (common.py:291-422)
// ---------------------------------------------------------------------------
EPSM_HD F3 xform_point(const float *m, F3 p) {                            // 4x4 row-major, with perspective divide
    const float x = m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], y = m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
                z = m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11], w = m[12] * p.x + m[13] * p.y + m[14] * p.z + m[15];
    const float iw = 1.f / w;
    return f3(x * iw, y * iw, z * iw);
}
EPSM_HD F3 xform_vec34(const float *m, F3 v) {
    return f3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z, m[8] * v.x + m[9] * v.y + m[10] * v.z);
}
struct PrimaryRay { Ray ray; F3 dx, dy; float px, py; };
// PerspectiveCamera::sample_ray_differential (perspective.cpp:238-279) at the film position (fx, fy) in pixels
EPSM_HD PrimaryRay primary_ray_at(const EpsmSensor &C, float fx, float fy) {
    PrimaryRay o;
    o.px = fx; o.py = fy;
    const float sx = o.px / C.width, sy = o.py / C.height;
    const F3 near_p = xform_point(C.sample_to_camera, f3(sx, sy, 0.f));   // perspective.cpp:255-258
    const F3 d = normalize3(near_p);
    const float *W = C.to_world;                                          // 3x4
    const F3 origin = f3(W[3], W[7], W[11]);
    const F3 dw = xform_vec34(W, d);
    const float inv_z = 1.f / d.z, near_t = C.near_clip * inv_z, far_t = C.far_clip * inv_z;
    o.ray.o = origin + dw * near_t;                                       // perspective.cpp:266-270
    o.ray.d = dw;
    o.ray.maxt = far_t - near_t;
    o.dx = xform_vec34(W, normalize3(near_p + ld3(C.dx)));               // perspective.cpp:274-275
    o.dy = xform_vec34(W, normalize3(near_p + ld3(C.dy)));
    return o;
}
EPSM_HD PrimaryRay sample_primary_ray(const EpsmSensor &C, int64_t wavefront_index, int spp, Pcg32 &rng) {
    // common.py:320-335: idx // spp -> pixel, pos = pixel + next_2d()
    const int64_t pix = wavefront_index / spp;
    const int fw = C.width + 2 * C.border;                                // film_size += 2 * border_size (common.py:314-315)
    const int py0 = (int) (pix / fw), px = (int) (pix - (int64_t) py0 * fw) - C.border, py = py0 - C.border;
    const float jx = rng.next_1d(), jy = rng.next_1d();
    return primary_ray_at(C, px + jx, py + jy);
}

// GaussianFilter::eval (src/rfilters/gaussian.cpp): stddev 0.5, cut off after 4 standard deviations, shifted so that
// it reaches 0 at the radius; the film kernels evaluate it exactly (no discretisation table)
constexpr float kGaussRadius = 2.f, kGaussAlpha = -1.f / (2.f * 0.5f * 0.5f);
EPSM_HD float gaussian_rfilter(float x) {
    const float bias = expf(kGaussAlpha * kGaussRadius * kGaussRadius);
    return fabsf(x) <= kGaussRadius ? fmaxf(0.f, expf(kGaussAlpha * x * x) - bias) : 0.f;
}

// ---------------------------------------------------------------------------
// the path (epsm.py:503-742)
// ---------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__) && defined(EPSM_TRACE_NT_STORES)
// the log is written once and read by a later kernel: streaming stores
template <class T> EPSM_HD void st1(T *p, T v) { __builtin_nontemporal_store(v, p); }
#else
template <class T> EPSM_HD void st1(T *p, T v) { *p = v; }
#endif
EPSM_HD void st3(float *base, int64_t i, F3 v) { if (base) { st1(base + 3 * i, v.x); st1(base + 3 * i + 1, v.y); st1(base + 3 * i + 2, v.z); } }
EPSM_HD uint32_t f2u(float f) { union { float f; uint32_t u; } c; c.f = f; return c.u; }

struct TraceArgs {
    EpsmScene S;
    EpsmSensor C;
    uint32_t seed;
    int spp, max_depth, rr_depth, K_log;
    uint32_t flags;                  // EPSM_TRACE_*
    int64_t path_offset, N;
    float *ray_o, *ray_d, *ray_dx, *ray_dy, *film_pos, *radiance;
    uint8_t *valid;
    EpsmRecordOut rec[kMaxVertices];
    float *color_sum;                // epsm_trace_paths_color: (N, n_color, 3), else null
    int n_color;
    int64_t pk_ray_stride, pk_stride; // EPSM_TRACE_PACKED_LOG: words between consecutive paths' rays / first records (EpsmRecordOut)
    EpsmFirstHitBackward fh;         // EPSM_TRACE_FUSE_FIRST_HIT: copied from recs[0].first_hit
};
// after rec[], flags and K_log are set: the strides of the native log (EpsmRecordOut.ray_stride / packed_stride; 0 = dense)
inline void trace_args_log_strides(TraceArgs &A) {
    const bool pk = (A.flags & EPSM_TRACE_PACKED_LOG) && A.K_log > 0;
    A.pk_ray_stride = (pk && A.rec[0].ray_stride) ? A.rec[0].ray_stride : 12;
    A.pk_stride = (pk && A.rec[0].packed_stride) ? A.rec[0].packed_stride : (int64_t) A.K_log * 32;
}
// record of bounce `iteration` of path i in the native log
EPSM_HD float *packed_record(const TraceArgs &A, int64_t i, int iteration) { return A.rec[0].packed + i * A.pk_stride + iteration * 32; }

EPSM_HD void write_record(const EpsmRecordOut &R, int64_t i, bool active, const SurfHit &h, uint32_t flags,
                          const EmitterSample &es, bool active_em, const BsdfSample &bs, float eweight, int32_t alpha_slot) {
    const bool mesh = h.valid && (h.mesh_flags & EPSM_MESH_IS_MESH);
    const F3 z = zero3<float>();
    // analytic shapes leave the EPSM fields zero and ismesh = 0 (interaction.h:221-224 zero-initialised)
    st3(R.p0, i, mesh ? h.p0 : z); st3(R.p1, i, mesh ? h.p1 : z); st3(R.p2, i, mesh ? h.p2 : z); st3(R.p, i, h.p);
    st3(R.n0, i, mesh ? h.n0 : z); st3(R.n1, i, mesh ? h.n1 : z); st3(R.n2, i, mesh ? h.n2 : z); st3(R.normal, i, h.shn);
    st1(R.b0 + i, mesh ? h.b0 : 0.f); st1(R.b1 + i, mesh ? h.b1 : 0.f);
    st1(R.eta + i, bs.eta);
    st3(R.hf, i, bs.hf); st3(R.light, i, es.p);
    st1(R.bsdf + i, flags);
    st1(R.active + i, (uint8_t) (active ? 1 : 0)); st1(R.active_em + i, (uint8_t) (active_em ? 1 : 0)); st1(R.ismesh + i, (uint8_t) (mesh ? 1 : 0));
    st1(R.tri + i, mesh ? h.tri : kNoIndex);                     // row of the scene's triangle table (include/epsm.h)
    uint32_t *a = R.aux + 4 * i;
    st1(a, alpha_slot >= 0 ? (uint32_t) alpha_slot : kNoIndex); st1(a + 1, f2u(bs.dhf.x)); st1(a + 2, f2u(bs.dhf.y)); st1(a + 3, f2u(bs.dhf.z));
    uint32_t *e = R.emit + 4 * i;
    st1(e, es.tri); st1(e + 1, f2u(es.b0)); st1(e + 2, f2u(es.b1)); st1(e + 3, f2u(eweight));
}

// EPSM_TRACE_PACKED_LOG: the same record as ONE 128-byte row of the native log (include/epsm.h, EpsmPackedLog): eight
// 16-byte stores instead of twenty scattered ones, plus the path's flag word so far (five bits per logged bounce).
EPSM_HD void st4(float *p, float a, float b, float c, float d) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef float F4s __attribute__((ext_vector_type(4)));
    F4s v; v.x = a; v.y = b; v.z = c; v.w = d;
#if defined(EPSM_TRACE_NT_STORES) && defined(EPSM_TRACE_NT_PACKED)
    __builtin_nontemporal_store(v, (F4s *) p);
#else
    *(F4s *) p = v;      // a lane writes the eight quads of its record back to back: left to L2 to combine into full lines
#endif
#else
    p[0] = a; p[1] = b; p[2] = c; p[3] = d;
#endif
}
EPSM_HD float u2f(uint32_t u) { union { uint32_t u; float f; } c; c.u = u; return c.f; }
// the five bits of a logged vertex in a path's flag word (include/epsm.h EPSM_FLAG_*): Diffuse, Null, active, active_em, mesh
EPSM_HD uint32_t vertex_flag_bits(bool active, const SurfHit &h, uint32_t flags, bool active_em) {
    const bool mesh = h.valid && (h.mesh_flags & EPSM_MESH_IS_MESH);
    return ((flags & 0x6u) ? 1u : 0u) | ((flags & 0x1u) ? 2u : 0u) | (active ? 4u : 0u) | (active_em ? 8u : 0u) | (mesh ? 16u : 0u);
}
// `word` = the path's flag word INCLUDING this vertex (PathState::gword)
EPSM_HD void write_record_packed(float *r, uint32_t *pflags, int64_t i, uint32_t word,
                                 const SurfHit &h, const EmitterSample &es,
                                 const BsdfSample &bs, float eweight) {
    const bool mesh = h.valid && (h.mesh_flags & EPSM_MESH_IS_MESH);
    const F3 z = zero3<float>();
    const F3 p0 = mesh ? h.p0 : z, p1 = mesh ? h.p1 : z, p2 = mesh ? h.p2 : z, n0 = mesh ? h.n0 : z, n1 = mesh ? h.n1 : z,
             n2 = mesh ? h.n2 : z;
    const float tid = u2f(mesh ? h.tri : kNoIndex);
    st4(r + 0, p0.x, p0.y, p0.z, p1.x);                                   // (layout: include/epsm.h, EpsmPackedLog)
    st4(r + 4, p1.y, p1.z, p2.x, p2.y);
    st4(r + 8, p2.z, mesh ? h.b0 : 0.f, mesh ? h.b1 : 0.f, tid);
    st4(r + 12, n0.x, n0.y, n0.z, bs.eta);
    st4(r + 16, n1.x, n1.y, n1.z, n2.x);
    st4(r + 20, n2.y, n2.z, es.p.x, es.p.y);
    st4(r + 24, u2f(es.tri), es.b0, es.b1, eweight);
    st4(r + 28, es.p.z, bs.dhf.x, bs.dhf.y, bs.dhf.z);
    pflags[i] = word;                                                     // (the path's lane is the word's only writer)
}
// The FIRST SECTOR of the record alone (geometry, barycentrics, triangle id, n0; eta = 0): all that is ever read of a vertex
// that is only looked at -- a chain's end point, a diffuse first hit (include/epsm.h, EpsmPackedLog)
EPSM_HD void write_record_packed_first_sector(float *r, uint32_t *pflags, int64_t i, uint32_t word, const SurfHit &h) {
    const bool mesh = h.valid && (h.mesh_flags & EPSM_MESH_IS_MESH);
    const F3 z = zero3<float>();
    const F3 p0 = mesh ? h.p0 : z, p1 = mesh ? h.p1 : z, p2 = mesh ? h.p2 : z, n0 = mesh ? h.n0 : z;
    st4(r + 0, p0.x, p0.y, p0.z, p1.x);
    st4(r + 4, p1.y, p1.z, p2.x, p2.y);
    st4(r + 8, p2.z, mesh ? h.b0 : 0.f, mesh ? h.b1 : 0.f, u2f(mesh ? h.tri : kNoIndex));
    st4(r + 12, n0.x, n0.y, n0.z, 0.f);
    pflags[i] = word;
}

// EPSM_TRACE_SPARSE_LOG: a bounce the path did not reach leaves only the fields the gradient kernels' masks read
EPSM_HD void write_dead_masks(const EpsmRecordOut &R, int64_t i) {
    st1(R.bsdf + i, 0u);
    st1(R.active + i, (uint8_t) 0); st1(R.active_em + i, (uint8_t) 0); st1(R.ismesh + i, (uint8_t) 0);
}
// Default record of the occluder term: "no occluder" (epsm.py:609-620 not taken)
EPSM_HD void write_no_occluder(uint32_t *o) {
    o[0] = kNoIndex; o[1] = o[2] = o[3] = 0u;
}
// Occluder of the first vertex's emitter sample (epsm.py:609-620; integrators with max_depth <= 3): closest hit of the
// ray towards the sample, NO maximum distance, as scene.ray_intersect(si.spawn_ray(ds.d)).  `sr` = spawn_ray(si, ds.d),
// `oh` its closest hit, sip = si.p, esp = ds.p.
EPSM_HD void write_occluder(const EpsmScene &S, uint32_t *o, const Ray &sr, const TriHit &oh, F3 sip, F3 esp) {
    uint32_t w[4] = {kNoIndex, 0u, 0u, 0u};
    if (oh.hit) {
        const SurfHit occ = surface_interaction(S, sr, oh);
        if (occ.mesh_flags & EPSM_MESH_IS_MESH) {
            const F3 a = esp - occ.p, b = esp - sip;
            float dis = sqrtf(dot(a, a)) / sqrtf(dot(b, b));                     // :614
            if (!(dis >= 0.01f)) dis = 0.f;                                      // :615
            w[0] = occ.tri; w[1] = f2u(occ.b0); w[2] = f2u(occ.b1); w[3] = f2u(dis);
        }
    }
    for (int j = 0; j < 4; ++j) o[j] = w[j];
}

// What a path carries from one bounce to the next (the loop state of epsm.py:527-545)
struct PathState {
    Ray ray;
    F3 L, beta, prev_p;
    float eta, prev_bsdf_pdf;
    int depth;
    bool active, prev_bsdf_delta;
    Pcg32 rng;
    uint32_t cnt;                    // colour adjoint: vertices passed so far per BSDF colour slot, 8 bits each
    uint32_t gword;                  // flag word of the vertices logged so far (five bits each, vertex_flag_bits)
};

// ---- EPSM_TRACE_FUSE_FIRST_HIT: the backward pass of a path without a chain, as csrc/epsm_backward_cp.hip does it for a lane of
// class 0 (epsm.py:250-272, 561-562, 791-792) -- the same functions on the same numbers.
struct FirstHitRows { bool on; uint32_t key[3]; F3 val[3]; };              // rows of grad_pos this path adds to
// d loss / d film position of path i's pixel (channels 3, 4 of the gradient image; epsm.py:250-255)
EPSM_HD void first_hit_pixel_grad(const TraceArgs &A, int64_t i, float &gx, float &gy) {
    // (32-bit divisions: the wavefront index fits 32 bits -- fill_trace_args, common.py:468-475 -- and two 64-bit ones are ~200 instructions)
    const uint32_t widx = (uint32_t) (A.path_offset + i), pix = widx / (uint32_t) A.spp, y = pix / (uint32_t) A.fh.res, x = pix - y * (uint32_t) A.fh.res;
    const float *g = A.fh.grad_img + ((int64_t) y * A.fh.img_width + x) * A.fh.img_channels;
    gx = g[3]; gy = g[4];
}
// `w`: the flag word (vertex 1 alone), `gd` = (d_x - d) gx + (d_y - d) gy, `ray`: the primary ray
EPSM_HD FirstHitRows first_hit_rows(const TraceArgs &A, uint32_t w, const SurfHit &si, const Ray &ray, F3 gd) {
    FirstHitRows r;
    r.on = false; r.key[0] = r.key[1] = r.key[2] = kNoIndex; r.val[0] = r.val[1] = r.val[2] = zero3<float>();
    const bool mesh = si.valid && (si.mesh_flags & EPSM_MESH_IS_MESH);
    const bool d1 = (w & 1u) != 0, act1 = (w & 4u) != 0;                    // cp::plan_diffuse1, cp::kPlanActive1
    if (!d1 || !mesh) return r;                                            // (no Diffuse lobe: no diffuse_grad[0]; no triangle: no rows)
    const F3 z = zero3<float>();
    const Tangent t = tangent_from_gd(ray.o, ray.d, gd, act1 ? si.p0 : z, act1 ? si.p1 : z, act1 ? si.p2 : z, act1);
    const U4 row = table_row(TriTable{A.fh.tri_table, A.fh.T}, si.tri);
    const bool ok = row.x < (uint64_t) A.fh.V && row.y < (uint64_t) A.fh.V && row.z < (uint64_t) A.fh.V && (row.w & kModePos);
    if (!ok) return r;
    const float clip = (A.fh.clip > 0.0f && A.fh.clip <= 3.402823466e+38f) ? A.fh.clip : 3.402823466e+38f;
    const F3 dpf = f3(finalize(t.dp.x, clip), finalize(t.dp.y, clip), finalize(t.dp.z, clip));
    const float b0 = si.b0, b1 = si.b1, b2 = 1.f - b0 - b1;
    r.key[0] = row.x; r.key[1] = row.y; r.key[2] = row.z;
    r.val[0] = dpf * b0; r.val[1] = dpf * b1; r.val[2] = dpf * b2;
    r.on = nz3(r.val[0]) || nz3(r.val[1]) || nz3(r.val[2]);
    return r;
}

// sample_rays (common.py:291-422) + the initial loop state
// `log = false`: the ray only, nothing written (the wavefront tracer's first closest-hit stage re-derives the primary ray
// instead of reading a stored one)
EPSM_HD PathState path_begin(const TraceArgs &A, int64_t i, bool log = true, PrimaryRay *pr_out = nullptr) {
    const int64_t widx = A.path_offset + i;
    PathState s;
    s.rng = seed_sampler(A.seed, (uint32_t) widx);                        // common.py:475 sampler.seed(seed, wavefront_size)
    const PrimaryRay pr = sample_primary_ray(A.C, widx, A.spp, s.rng);
    if (pr_out) *pr_out = pr;
    if (!log) {
    } else if (A.flags & EPSM_TRACE_PACKED_LOG) {                         // (N,12): o, d, d_x, d_y side by side
        float *r = A.ray_o + A.pk_ray_stride * i;
        st4(r, pr.ray.o.x, pr.ray.o.y, pr.ray.o.z, pr.ray.d.x);
        st4(r + 4, pr.ray.d.y, pr.ray.d.z, pr.dx.x, pr.dx.y);
        st4(r + 8, pr.dx.z, pr.dy.x, pr.dy.y, pr.dy.z);
        // (interleaved block, ray stride >= 16: writing the sector's four free words too, so that no 48-byte store needs a
        // read-modify-write at the memory side, measured SLOWER -- trace + log of 2^24 paths 3.63 -> 3.82 ms, the dense arrays 3.45)
    } else {
        st3(A.ray_o, i, pr.ray.o); st3(A.ray_d, i, pr.ray.d); st3(A.ray_dx, i, pr.dx); st3(A.ray_dy, i, pr.dy);
    }
    if (log && A.film_pos) { A.film_pos[2 * i] = pr.px; A.film_pos[2 * i + 1] = pr.py; }
    s.ray = pr.ray;
    s.L = zero3<float>(); s.beta = f3(1.f, 1.f, 1.f); s.prev_p = zero3<float>();
    s.eta = 1.f; s.prev_bsdf_pdf = 1.f;
    s.depth = 0;
    s.active = true; s.prev_bsdf_delta = true;
    s.cnt = 0u;
    s.gword = 0u;
    return s;
}
EPSM_HD int path_max_depth(const TraceArgs &A) { return A.max_depth < 6 ? A.max_depth : 6; }   // epsm.py:549

// One iteration of the loop of epsm.py:551-735 AFTER the closest hit `th` of s.ray is known.  The two places
// where the reference traces further rays from inside the iteration go through the policy `vis`:
//   vis.occluded(S, ray)                      visibility of the emitter sample (scene.cpp:270-275)
//   vis.direct(L, Le, Lr_dir)                 L += Le + Lr_dir (epsm.py:658); a deferred policy adds Lr_dir once it knows
//   vis.occluder(A, i, si, es, active_em)     the occluder record of the first vertex (epsm.py:609-620)
// so that the one-launch tracer answers them on the spot (InlineVis) and the wavefront tracer
// (epsm_trace_wavefront.h) queues them for its shadow-ray stage.
// `obs.vertex(...)` sees what the bounce computed before the state moves on (the reparameterised backward pass,
// epsm_trace_reparam.h, replays paths through this very function); the default observer is empty.
struct NoObserver {
    EPSM_HD void vertex(const SurfHit &, const EpsmBsdf &, uint32_t, F3, F3, const EmitterSample &, bool, float, const BsdfSample &, bool) {}
};
// EPSM_TRACE_FUSE_FIRST_HIT: does the rule retire a path at its FIRST vertex, hit `th`?  Follows from the mesh's and the BSDF's flag
// bits alone; `lite` then holds what first_hit_rows reads -- validity, mesh flags, triangle id and, for a diffuse mesh hit, the
// triangle's positions and the barycentrics -- and `w` the vertex's five flag bits.
EPSM_HD bool first_hit_retires(const TraceArgs &A, const TriHit &th, uint32_t &w, SurfHit &lite) {
    const EpsmScene &S = A.S;
    uint32_t mflags = 0, bflags = 0;
    if (th.hit) {
        const EpsmMesh m = S.meshes[S.tri_mesh[th.tri]];
        mflags = m.flags;
        if (m.bsdf >= 0) bflags = bsdf_flags(S.bsdfs[m.bsdf]);
    }
    lite.valid = th.hit; lite.mesh_flags = mflags; lite.tri = th.tri;
    w = vertex_flag_bits(th.hit, lite, bflags, false);
    if (cp::gradient_live(w, 1, (A.flags & EPSM_TRACE_GRADIENT_CAUSTIC) != 0)) return false;
    lite.p0 = lite.p1 = lite.p2 = zero3<float>();
    lite.b0 = lite.b1 = lite.b2 = 0.f;
    if (th.hit && (w & 1u) && (mflags & EPSM_MESH_IS_MESH)) {              // (rows exist for a diffuse mesh hit only: first_hit_rows)
        const uint32_t *iv = S.tri + 3 * (int64_t) th.tri;
        lite.p0 = ld3(S.positions + 3 * (int64_t) iv[0]);
        lite.p1 = ld3(S.positions + 3 * (int64_t) iv[1]);
        lite.p2 = ld3(S.positions + 3 * (int64_t) iv[2]);
        lite.b1 = th.u; lite.b2 = th.v; lite.b0 = 1.f - th.u - th.v;
    }
    return true;
}
template <class Vis, class Obs>
EPSM_HD void path_bounce(const TraceArgs &A, int64_t i, int iteration, PathState &s, const TriHit &th, Vis &vis, Obs &obs) {
    const EpsmScene &S = A.S;
    // ---- EPSM_TRACE_FUSE_FIRST_HIT, first bounce: a path the rule retires here needs no normals, no texture coordinates, no shading
    // frame (94 % of the clutter scene's paths: the shade stage 0.94 -> 0.89 ms; the wavefront's packet stage does the same for its
    // primary rays and such a path never reaches this function)
    if (iteration == 0 && (A.flags & EPSM_TRACE_FUSE_FIRST_HIT) && s.active && A.K_log > 0) {
        uint32_t w;
        SurfHit lite;
        if (first_hit_retires(A, th, w, lite) && vis.first_hit(A, i, w, lite, s.ray)) {
            s.gword = w;
            if (th.hit) s.depth += 1;
            s.active = false;
            return;
        }
    }
    const SurfHit si = surface_interaction(S, s.ray, th);                 // epsm.py:556-558
    EpsmBsdf bsdf;
    bsdf.type = EPSM_BSDF_DIFFUSE_T; bsdf.twosided = 0; bsdf.distr = 0; bsdf.sample_visible = 1; bsdf.alpha = 0.1f;
    bsdf.reflectance[0] = bsdf.reflectance[1] = bsdf.reflectance[2] = 0.f;
    bsdf.eta[0] = bsdf.eta[1] = bsdf.eta[2] = 0.f; bsdf.k[0] = bsdf.k[1] = bsdf.k[2] = 1.f;
    bsdf.int_ior = 1.5046f; bsdf.ext_ior = 1.000277f; bsdf.alpha_slot = -1; bsdf.color_slot = -1; bsdf.texture = -1; bsdf.pad = 0;
    uint32_t flags = 0;
    if (si.valid && si.bsdf >= 0) { bsdf = S.bsdfs[si.bsdf]; flags = bsdf_flags(bsdf); }
    if (si.valid && bsdf.texture >= 0 && bsdf.texture < S.n_textures) {   // the reflectance at this point: everything below sees it as a constant
        const F3 r = tex_eval(S.textures[bsdf.texture], si.uvx, si.uvy);
        bsdf.reflectance[0] = r.x; bsdf.reflectance[1] = r.y; bsdf.reflectance[2] = r.z;
    }

    // ---- EPSM_TRACE_GRADIENT_ONLY, native log: a vertex at which the path is retired BY THE RULE (cp::gradient_live: a Diffuse,
    // non-mesh or missed vertex -- not the K_log limit) is only ever looked at: a chain's end point or a diffuse first hit, of
    // which the backward kernel reads the first sector of the record and the Diffuse / Null / active / mesh bits.  No emitter
    // sample, no BSDF sample, no second sector.  (Not when the occluder record of vertex 1 is wanted: it rides on the emitter sample.)
    if ((A.flags & EPSM_TRACE_GRADIENT_ONLY) && (A.flags & EPSM_TRACE_PACKED_LOG) && iteration < A.K_log && s.active &&
        !(iteration == 0 && A.rec[0].shadow && A.max_depth <= 3)) {
        const uint32_t w = s.gword | (vertex_flag_bits(si.valid, si, flags, false) << (5 * iteration));
        if (!cp::gradient_live(w, iteration + 1, (A.flags & EPSM_TRACE_GRADIENT_CAUSTIC) != 0)) {
            s.gword = w;
            // EPSM_TRACE_FUSE_FIRST_HIT: a path retired at its FIRST vertex has no chain; what the backward pass would do for it
            // (first_hit_rows below) the shading stage does, and the log holds nothing of it (flag word 0)
            if (!(iteration == 0 && (A.flags & EPSM_TRACE_FUSE_FIRST_HIT) && vis.first_hit(A, i, w, si, s.ray)))
                write_record_packed_first_sector(packed_record(A, i, iteration), A.rec[0].pflags, i, w, si);
            if (si.valid) s.depth += 1;
            s.active = false;
            return;
        }
    }
    // ---- direct emission, MIS against the emitter sample of the previous bounce (epsm.py:569-577)
    F3 Le = zero3<float>();
    if (si.valid && si.emitter >= 0 && si.wi.z > 0.f) {                    // area.cpp eval: front side only
        const float em_pdf = s.prev_bsdf_delta ? 0.f : pdf_emitter_direction(S, s.prev_p, si);
        const float mis = mis_weight(s.prev_bsdf_pdf, em_pdf);
        Le = mul3(s.beta, ld3(S.emitters[si.emitter].radiance)) * mis;
    } else if (!si.valid && has_environment(S)) {                          // the ray left the scene: the environment's radiance
        float em_pdf = s.prev_bsdf_delta ? 0.f : env_pdf(S, s.ray.d);
        if (S.n_emitters > 1) em_pdf /= (float) S.n_emitters;
        Le = mul3(s.beta, env_eval(S, s.ray.d)) * mis_weight(s.prev_bsdf_pdf, em_pdf);
    }
    // ---- emitter sampling (epsm.py:582-605)
    bool active_next = (s.depth + 1 < A.max_depth) && si.valid;
    bool active_em = active_next && (flags & kFlSmooth);
    const float e1 = s.rng.next_1d(), e2 = s.rng.next_1d();              // sampler.next_2d()
    const EmitterSample es = sample_emitter_direction(S, si, e1, e2, active_em, vis, false);
    active_em = active_em && es.pdf != 0.f;                               // :590
    F3 Lr_dir = zero3<float>();
    float mis_em = 0.f;
    if (active_em) {
        const F3 wo = to_local(si, es.d);
        F3 bval; float bpdf;
        bsdf_eval_pdf(bsdf, si.wi, wo, bval, bpdf);
        mis_em = es.delta ? 1.f : mis_weight(es.pdf, bpdf);
        Lr_dir = mul3(mul3(s.beta, bval), es.weight) * mis_em;            // :605
        // The visibility ray of scene.cpp:270-275, AFTER the BSDF value is known: an occluded sample only zeroes
        // ds.weight, i.e. Lr_dir and the logged emitter weight -- when they are zero already (emitter below the
        // surface's horizon: half of the samples on a convex object) the ray decides nothing and is not traced.
        // (The occluder record of the first vertex rides on this ray in the wavefront form: always traced then.)
        const bool occluder_wanted = iteration == 0 && A.K_log > 0 && A.rec[0].shadow && A.max_depth <= 3;
        // EPSM_TRACE_GRADIENT_ONLY: the ray decides Lr_dir, i.e. the logged weight eweight = sum Lr_dir, which calc_grad reads
        // in ONE place -- light_grad x eweight of a light-sampling term wN(k) (epsm.py:622-627), which needs hasdiffuse == 0 and
        // every vertex through k a mesh hit (the condition under which a `manifold` path goes on), and which is identically zero
        // for manifold_caustic (csrc/epsm_path_core.h).  Elsewhere the weight is never looked at: no ray.
        bool weight_read = true;
        if (A.flags & EPSM_TRACE_GRADIENT_ONLY) {
            const uint32_t w = s.gword | (vertex_flag_bits(s.active && si.valid, si, flags, active_em) << (5 * iteration));
            weight_read = !(A.flags & EPSM_TRACE_GRADIENT_CAUSTIC) && iteration < A.K_log && cp::gradient_live(w, iteration + 1, false);
        }
        if (es.valid && ((weight_read && (Lr_dir.x != 0.f || Lr_dir.y != 0.f || Lr_dir.z != 0.f)) || occluder_wanted)) {
            float dist;
            const Ray sr = spawn_ray_to(si, es.p, dist);
            if (vis.occluded(S, sr)) Lr_dir = zero3<float>();
        }
    }
    // ---- occluder of the first vertex's emitter sample (epsm.py:609-620)
    if (iteration == 0 && A.K_log > 0 && A.rec[0].shadow) vis.occluder(A, i, si, es, active_em);
    // ---- BSDF sampling: once detached, once attached with fresh numbers (epsm.py:633-643)
    s.rng.next_1d(); s.rng.next_1d(); s.rng.next_1d();
    const float s1 = s.rng.next_1d(), s2x = s.rng.next_1d(), s2y = s.rng.next_1d();
    const BsdfSample bs = bsdf_sample(bsdf, si.wi, s1, s2x, s2y, active_next);
    // ---- log (epsm.py:648-654)
    if (iteration < A.K_log) {
        if (s.active || iteration == 0) s.gword |= vertex_flag_bits(s.active && si.valid, si, flags, active_em) << (5 * iteration);
        if (A.flags & EPSM_TRACE_PACKED_LOG) {
            if (s.active || iteration == 0)                                // (every path passes bounce 0: its flag word exists)
                write_record_packed(packed_record(A, i, iteration), A.rec[0].pflags, i, s.gword, si, es, bs,
                                    Lr_dir.x + Lr_dir.y + Lr_dir.z);
        } else if (s.active || !(A.flags & EPSM_TRACE_SPARSE_LOG)) {
            write_record(A.rec[iteration], i, s.active && si.valid, si, flags, es, active_em, bs,
                         Lr_dir.x + Lr_dir.y + Lr_dir.z, bsdf.alpha_slot);
        } else {
            write_dead_masks(A.rec[iteration], i);
        }
    }
    // ---- colour adjoint (epsm_trace_paths_color; needs the final Lr_dir, i.e. the one-launch form): this path's row of
    //      color_sum gets, per BSDF slot j, (vertices of j the term passed) x term, and per emitter slot the term itself
    if (A.color_sum && s.active) {
        float *G = A.color_sum + i * A.n_color * 3;
        const int here = (si.valid && bsdf.color_slot >= 0 && bsdf.color_slot < A.n_color) ? bsdf.color_slot : -1;
        for (int j = 0; j < A.n_color; ++j) {
            const float n = (float) ((s.cnt >> (8 * j)) & 0xFFu);
            const float nd = n + (j == here ? 1.f : 0.f);                  // the emitter-sample term includes this vertex's BSDF
            G[3 * j] += n * Le.x + nd * Lr_dir.x; G[3 * j + 1] += n * Le.y + nd * Lr_dir.y; G[3 * j + 2] += n * Le.z + nd * Lr_dir.z;
        }
        if (si.valid && si.emitter >= 0) {
            const int se = S.emitters[si.emitter].color_slot;
            if (se >= 0 && se < A.n_color) { G[3 * se] += Le.x; G[3 * se + 1] += Le.y; G[3 * se + 2] += Le.z; }
        } else if (!si.valid && has_environment(S)) {                     // the ray left the scene: Le is the environment's
            const int se = S.emitters[S.env.emitter].color_slot;
            if (se >= 0 && se < A.n_color) { G[3 * se] += Le.x; G[3 * se + 1] += Le.y; G[3 * se + 2] += Le.z; }
        }
        if (es.emitter >= 0) {
            const int se = S.emitters[es.emitter].color_slot;
            if (se >= 0 && se < A.n_color) { G[3 * se] += Lr_dir.x; G[3 * se + 1] += Lr_dir.y; G[3 * se + 2] += Lr_dir.z; }
        }
        if (here >= 0 && bs.valid) s.cnt += 1u << (8 * here);              // the sampled direction carries one more factor rho_j
    }
    obs.vertex(si, bsdf, flags, Le, Lr_dir, es, active_em, mis_em, bs, s.active);
    // ---- update (epsm.py:658-683)
    if (s.active) vis.direct(s.L, Le, Lr_dir);
    const F3 wo_world = to_world(si, bs.wo);
    s.ray = spawn_ray(si, wo_world);
    s.eta *= bs.valid ? bs.eta : 1.f;
    s.beta = bs.valid ? mul3(s.beta, bs.weight) : zero3<float>();
    s.prev_p = si.p;
    s.prev_bsdf_pdf = bs.pdf;
    s.prev_bsdf_delta = (bs.sampled_type & kFlDelta) != 0;
    const float beta_max = max3(s.beta);
    active_next = active_next && beta_max != 0.f;
    const float rr_prob = fminf(beta_max * s.eta * s.eta, 0.95f);
    const bool rr_active = s.depth >= A.rr_depth;
    if (rr_active) s.beta = s.beta * (1.f / rr_prob);
    const bool rr_continue = s.rng.next_1d() < rr_prob;
    active_next = active_next && (!rr_active || rr_continue);
    if (si.valid && s.active) s.depth += 1;                               // :734
    s.active = s.active && active_next;                                   // :735
    // EPSM_TRACE_GRADIENT_ONLY (include/epsm_trace.h): nothing behind this vertex can reach calc_grad -> the path ends here
    if ((A.flags & EPSM_TRACE_GRADIENT_ONLY) &&
        (iteration + 1 >= A.K_log || !cp::gradient_live(s.gword, iteration + 1, (A.flags & EPSM_TRACE_GRADIENT_CAUSTIC) != 0)))
        s.active = false;
}
template <class Vis>
EPSM_HD void path_bounce(const TraceArgs &A, int64_t i, int iteration, PathState &s, const TriHit &th, Vis &vis) {
    NoObserver obs;
    path_bounce(A, i, iteration, s, th, vis, obs);
}
EPSM_HD void path_end(const TraceArgs &A, int64_t i, const PathState &s) {
    st3(A.radiance, i, s.L);
    if (A.valid) A.valid[i] = s.depth != 0;
}

// Every further ray answered on the spot, with the path's own traversal stack.
struct InlineVis {
    const BvhStack &st;
    EPSM_HD bool occluded(const EpsmScene &S, const Ray &sr) { return intersect<true>(S, sr, st).hit; }
    EPSM_HD void direct(F3 &L, F3 Le, F3 Lr_dir) { L = L + Le + Lr_dir; }
    EPSM_HD bool first_hit(const TraceArgs &, int64_t, uint32_t, const SurfHit &, const Ray &) { return false; }   // (the wavefront's shade stage only)
    EPSM_HD void occluder(const TraceArgs &A, int64_t i, const SurfHit &si, const EmitterSample &es, bool active_em) {
        uint32_t *o = A.rec[0].shadow + 4 * i;
        if (A.max_depth <= 3 && active_em) {
            const Ray sr = spawn_ray(si, es.d);
            write_occluder(A.S, o, sr, intersect<false>(A.S, sr, st), si.p, es.p);
        } else {
            write_no_occluder(o);
        }
    }
};

// The whole path in one go (one lane carries it through all bounces; lanes stay in the loop, masked: epsm.py:551).
EPSM_HD void trace_one_path(const TraceArgs &A, int64_t i, const BvhStack &st) {
    PathState s = path_begin(A, i);
    InlineVis vis{st};
    const int max_depth = path_max_depth(A);
    for (int iteration = 0; iteration < max_depth; ++iteration) {
        TriHit th; th.hit = false; th.tri = 0; th.t = kInf; th.u = th.v = 0.f;
        if (s.active) th = intersect<false>(A.S, s.ray, st);
        path_bounce(A, i, iteration, s, th, vis);                         // bounces never reached are logged as inactive zeros
    }
    path_end(A, i, s);
}

}  // namespace epsm
