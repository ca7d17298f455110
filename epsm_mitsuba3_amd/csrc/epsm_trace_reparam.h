// epsm_trace_reparam.h -- the reparameterised backward pass of the hybrid scheme's second phase:
// gradients of vertex positions / normals through visibility, i.e. what `prb_reparam` adds to path replay.
//
// Restates, for the estimator of epsm_trace_core.h (path_bounce) and triangle meshes,
//   _sample_warp_field / _ReparameterizeOp.backward        src/python/python/ad/reparam.py:10-123, 224-333
//   PRBReparamIntegrator.sample (ADMode.Backward)           src/python/python/ad/integrators/prb_reparam.py:277-607
//   ADIntegrator.sample_rays (reparameterised film position) src/python/python/ad/integrators/common.py:376-421
//   Mesh::compute_surface_interaction's three AD modes + boundary test   src/render/mesh.cpp:652-700, 832-887
//   Rectangle's boundary test                                src/shapes/rectangle.cpp:320-321
// WITHOUT an AD system: a path replays under the primal pass's seed; every vertex `cur` evaluates its contribution
//   Lo = (Le + Lr_dir + Lr_ind) * det + extra                (prb_reparam.py:572)
// once more in DUAL numbers (forward mode) with respect to the few local quantities the reference's AD graph reaches
// from there -- the reparameterised direction d' of the ray into the vertex, that of the emitter ray, the three vertices
// and vertex normals of the triangle -- and hands the resulting adjoints (a) straight to the triangle's rows of the
// gradient buffers and (b) to the hand-derived adjoint of the warp field (reparam.py:269-327), which distributes them over
// the triangles its auxiliary rays hit (`FollowShape`: si.p = sum b_j p_j with the b_j detached) and over the ray origin.
// Sampling is detached as in the reference (pdfs, MIS weights, emitter samples are constants).
// Where a warp is traced is a policy (`Sink`): on the spot, one auxiliary ray after the other (InlineSink: the host build,
// the test probes), or left as a request for a second launch that gives every auxiliary ray its own lane (QueueSink:
// epsm_trace_reparam.hip).  Every auxiliary ray has its own random stream, so both trace the same rays.
//
// PARITY UNPINNED by the reference (no Dr.Jit here): pinned by the reference's OWN recipe for this integrator
// (src/integrators/tests/test_ad_integrators.py:833-871: backward gradient against finite differences of the primal
// image), tests/test_reparam*.py, on the host build of this file and on the GPU.
#pragma once

#include "epsm_trace_core.h"

namespace epsm {
namespace rp {

constexpr int kMaxAux = 64;              // reparam_rays <= 64 (the reference's tests use 64, its default is 16)
constexpr int kD = 6;                    // partials carried per dual evaluation

// ---------------------------------------------------------------------------
// dual numbers: value + kD partial derivatives
// ---------------------------------------------------------------------------
struct Dual {
    float v;
    float d[kD];
    EPSM_HD Dual() {}
    EPSM_HD Dual(float x) : v(x) {
#pragma unroll
        for (int i = 0; i < kD; ++i) d[i] = 0.f;
    }
};
EPSM_HD Dual operator+(Dual a, Dual b) { Dual r; r.v = a.v + b.v; for (int i = 0; i < kD; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
EPSM_HD Dual operator-(Dual a, Dual b) { Dual r; r.v = a.v - b.v; for (int i = 0; i < kD; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
EPSM_HD Dual operator-(Dual a) { Dual r; r.v = -a.v; for (int i = 0; i < kD; ++i) r.d[i] = -a.d[i]; return r; }
EPSM_HD Dual operator*(Dual a, Dual b) { Dual r; r.v = a.v * b.v; for (int i = 0; i < kD; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
EPSM_HD Dual operator/(Dual a, Dual b) {
    Dual r; const float ib = 1.f / b.v; r.v = a.v * ib;
    for (int i = 0; i < kD; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * ib;
    return r;
}
EPSM_HD Dual t_sqrt(Dual a) { Dual r; r.v = sqrtf(a.v); const float k = r.v > 0.f ? 0.5f / r.v : 0.f; for (int i = 0; i < kD; ++i) r.d[i] = a.d[i] * k; return r; }
EPSM_HD Dual t_exp(Dual a) { Dual r; r.v = expf(a.v); for (int i = 0; i < kD; ++i) r.d[i] = a.d[i] * r.v; return r; }
EPSM_HD float t_sqrt(float a) { return sqrtf(a); }
EPSM_HD float t_exp(float a) { return expf(a); }
EPSM_HD float val(float a) { return a; }
EPSM_HD float val(Dual a) { return a.v; }
// dr.replace_grad(primal, x): x's derivative on the stored primal value
EPSM_HD Dual replace_value(Dual a, float primal) { a.v = primal; return a; }
EPSM_HD float replace_value(float, float primal) { return primal; }

template <class T> EPSM_HD V3<T> lift3(F3 a) { return mk3<T>(T(a.x), T(a.y), T(a.z)); }
template <class T> EPSM_HD V3<T> t_normalize(V3<T> v) { const T il = T(1.f) / t_sqrt(dot(v, v)); return v * il; }
template <class T> EPSM_HD T t_mulsign(T x, T s) { return val(s) < 0.f ? -x : x; }
template <class T> EPSM_HD T t_sqr(T x) { return x * x; }
template <class T> EPSM_HD void t_coordinate_system(V3<T> n, V3<T> &s, V3<T> &t) {                 // vector.h (Duff et al.)
    const T sign = T(val(n.z) >= 0.f ? 1.f : -1.f), a = T(-1.f) / (sign + n.z), b = n.x * n.y * a;
    s = mk3<T>(t_mulsign(n.x * n.x * a, n.z) + T(1.f), t_mulsign(b, n.z), t_mulsign(-n.x, n.z));
    t = mk3<T>(b, n.y * n.y * a + sign, -n.y);
}

// ---------------------------------------------------------------------------
// the surface interaction as a function of (ray, triangle) -- mesh.cpp:652-827
// ---------------------------------------------------------------------------
template <class T> struct SurfT { V3<T> p, n, shn, fs, ft, wi; T t, bu, bv; };      // bu, bv: the weights of p1, p2 (prim_uv)
template <class T> EPSM_HD V3<T> to_local_t(const SurfT<T> &h, V3<T> v) { return mk3<T>(dot(v, h.fs), dot(v, h.ft), dot(v, h.shn)); }

// `u0, v0, t0`: the primal hit (prim_uv, t of the preliminary intersection: the derivative rides on them, mesh.cpp:690-695)
template <class T>
EPSM_HD SurfT<T> surf_t(V3<T> o, V3<T> d, V3<T> P0, V3<T> P1, V3<T> P2, V3<T> N0, V3<T> N1, V3<T> N2, uint32_t mesh_flags,
                        float u0, float v0, float t0) {
    SurfT<T> h;
    const V3<T> e1 = P1 - P0, e2 = P2 - P0;
    const V3<T> pvec = cross(d, e2);
    const T inv_det = T(1.f) / dot(e1, pvec);
    const V3<T> tvec = o - P0;
    const T u = replace_value(dot(tvec, pvec) * inv_det, u0);
    const V3<T> qvec = cross(tvec, e1);
    const T v = replace_value(dot(d, qvec) * inv_det, v0);
    h.t = replace_value(dot(e2, qvec) * inv_det, t0);
    const T b0 = T(1.f) - u - v;
    h.bu = u; h.bv = v;
    h.p = P0 * b0 + P1 * u + P2 * v;                                      // mesh.cpp:709
    h.n = t_normalize(cross(e1, e2));                                     // :729
    if (mesh_flags & EPSM_MESH_VERTEX_NORMALS) h.shn = t_normalize(N0 * b0 + N1 * u + N2 * v);     // :784-790
    else h.shn = h.n;
    if (mesh_flags & EPSM_MESH_FLIP_NORMALS) { h.n = -h.n; h.shn = -h.shn; }
    V3<T> dpdu, dpdv;
    t_coordinate_system(h.n, dpdu, dpdv);                                 // :734, SurfaceInteraction::initialize_sh_frame
    h.fs = t_normalize(dpdu - h.shn * dot(h.shn, dpdu));
    h.ft = cross(h.shn, h.fs);
    h.wi = to_local_t(h, -d);
    return h;
}

// ---------------------------------------------------------------------------
// BSDF values (incl. cosine) in T: diffuse.cpp:152-190, roughconductor.cpp:302-400; delta lobes evaluate to zero
// ---------------------------------------------------------------------------
template <class T> EPSM_HD T mf_eval_t(const EpsmBsdf &b, V3<T> m) {
    const float a = b.alpha;
    const T ct = m.z, ct2 = ct * ct;
    T result;
    if (b.distr == EPSM_DISTR_BECKMANN)
        result = t_exp(-(t_sqr(m.x * T(1.f / a)) + t_sqr(m.y * T(1.f / a))) / ct2) / (T(kPi * a * a) * t_sqr(ct2));
    else
        result = T(1.f) / (T(kPi * a * a) * t_sqr(t_sqr(m.x * T(1.f / a)) + t_sqr(m.y * T(1.f / a)) + t_sqr(m.z)));
    return val(result) * val(ct) > 1e-20f ? result : T(0.f);
}
template <class T> EPSM_HD T mf_smith_g1_t(const EpsmBsdf &b, V3<T> v, V3<T> m) {
    const T xy_alpha_2 = t_sqr(v.x * T(b.alpha)) + t_sqr(v.y * T(b.alpha)), tan2 = xy_alpha_2 / t_sqr(v.z);
    T result;
    if (b.distr == EPSM_DISTR_BECKMANN) {
        const T a = T(1.f) / t_sqrt(tan2), a2 = a * a;
        result = val(a) >= 1.6f ? T(1.f) : (a * T(3.535f) + a2 * T(2.181f)) / (T(1.f) + a * T(2.276f) + a2 * T(2.577f));
    } else {
        result = T(2.f) / (T(1.f) + t_sqrt(T(1.f) + tan2));
    }
    if (val(xy_alpha_2) == 0.f) result = T(1.f);
    if (val(dot(v, m)) * val(v.z) <= 0.f) result = T(0.f);
    return result;
}
template <class T> EPSM_HD T fresnel_conductor_t(T cos_i, float eta_r, float eta_i) {               // fresnel.h:92-117
    const T c2 = cos_i * cos_i, s2 = T(1.f) - c2, s4 = s2 * s2;
    const T temp_1 = T(eta_r * eta_r - eta_i * eta_i) - s2;
    const T x = temp_1 * temp_1 + T(4.f * eta_i * eta_i * eta_r * eta_r);
    const T a_2_pb_2 = val(x) > 0.f ? t_sqrt(x) : T(0.f);
    const T y = (a_2_pb_2 + temp_1) * T(0.5f);
    const T a = val(y) > 0.f ? t_sqrt(y) : T(0.f);
    const T term_1 = a_2_pb_2 + c2, term_2 = cos_i * a * T(2.f);
    const T r_s = (term_1 - term_2) / (term_1 + term_2);
    const T term_3 = a_2_pb_2 * c2 + s4, term_4 = term_2 * s2;
    const T r_p = r_s * (term_3 - term_4) / (term_3 + term_4);
    return (r_s + r_p) * T(0.5f);
}
template <class T> EPSM_HD V3<T> bsdf_eval_t(const EpsmBsdf &b, V3<T> wi, V3<T> wo) {
    if (b.twosided && val(wi.z) < 0.f) { wi.z = -wi.z; wo.z = -wo.z; }
    const T cti = wi.z, cto = wo.z;
    if (!(val(cti) > 0.f && val(cto) > 0.f)) return zero3<T>();
    if (b.type == EPSM_BSDF_DIFFUSE_T) return lift3<T>(ld3(b.reflectance)) * (cto * T(kInvPi));
    if (b.type == EPSM_BSDF_ROUGHCONDUCTOR_T) {
        const V3<T> H = t_normalize(wi + wo);
        const T D = mf_eval_t(b, H);
        if (val(D) == 0.f) return zero3<T>();
        const T G = mf_smith_g1_t(b, wi, H) * mf_smith_g1_t(b, wo, H);
        const T result = D * G / (cti * T(4.f));
        const T c = dot(wi, H);
        return mk3<T>(fresnel_conductor_t(c, b.eta[0], b.k[0]) * T(b.reflectance[0]) * result,
                      fresnel_conductor_t(c, b.eta[1], b.k[1]) * T(b.reflectance[1]) * result,
                      fresnel_conductor_t(c, b.eta[2], b.k[2]) * T(b.reflectance[2]) * result);
    }
    return zero3<T>();                                                    // conductor / dielectric: delta lobes
}

// ---------------------------------------------------------------------------
// the warp field (reparam.py:10-123): auxiliary rays around a ray, harmonic weights, the field's value and the
// ingredients of its divergence
// ---------------------------------------------------------------------------
struct Aux {
    float w;                 // harmonic weight (detached)
    F3 dw;                   // its gradient w.r.t. tangential changes of ray.d (detached)
    F3 v;                    // V_direct: normalize(si.p - ray.o), or ray.d for a miss
    uint32_t tri;            // triangle hit (kNoIndex: none)
    float b1, b2, inv_dist;  // barycentrics of the hit (FollowShape: detached), 1 / |si.p - ray.o|
};
struct Warp {
    int n;                   // auxiliary rays collected (0: reparameterisation inactive -> identity, no gradient)
    float Z;
    F3 dZ, o, d;
    Aux a[kMaxAux];
};
struct ReparamCfg { int max_depth, rays; float kappa, exponent; uint32_t flags; };      // flags: EPSM_REPARAM_* (epsm_trace.h)

// Boundary term of a hit (mesh.cpp:832-887; rectangle.cpp:320-321 for the tessellated rectangles)
EPSM_HD float boundary_test(const EpsmScene &S, const TriHit &th, F3 ray_d) {
    const uint32_t mi = S.tri_mesh[th.tri];
    const EpsmMesh m = S.meshes[mi];
    const float u = th.u, v = th.v, w = 1.f - u - v;
    if (!(m.flags & EPSM_MESH_IS_MESH)) {
        // rectangle = its first two triangles (v0 v1 v2), (v0 v2 v3): uv in [0,1]^2 along v1 - v0 and v2 - v1
        const uint32_t *i0 = S.tri + 3 * (int64_t) m.tri_begin;
        const F3 q0 = ld3(S.positions + 3 * (int64_t) i0[0]), q1 = ld3(S.positions + 3 * (int64_t) i0[1]), q2 = ld3(S.positions + 3 * (int64_t) i0[2]);
        const uint32_t *iv = S.tri + 3 * (int64_t) th.tri;
        const F3 p = ld3(S.positions + 3 * (int64_t) iv[0]) * w + ld3(S.positions + 3 * (int64_t) iv[1]) * u + ld3(S.positions + 3 * (int64_t) iv[2]) * v;
        const F3 eu = q1 - q0, ev = q2 - q1, r = p - q0;
        const float uu = dot(r, eu) / dot(eu, eu), vv = dot(r, ev) / dot(ev, ev);
        return fminf(0.5f - fabsf(uu - 0.5f), 0.5f - fabsf(vv - 0.5f));
    }
    const uint32_t *iv = S.tri + 3 * (int64_t) th.tri;
    if (!(m.flags & EPSM_MESH_VERTEX_NORMALS)) {
        // flat shading: distance to the nearest edge in an equilateral parameterisation, 1 at the barycentre
        const float s3 = 1.7320508075688772f;
        const float px = u + 0.5f * v, py = 0.5f * s3 * v;                // tp0 = (0,0), tp1 = (1,0), tp2 = (1/2, sqrt(3)/2)
        const float ex[3] = {1.f, -0.5f, -0.5f}, ey[3] = {0.f, 0.5f * s3, -0.5f * s3};
        const float ax[3] = {0.f, 1.f, 0.5f}, ay[3] = {0.f, 0.f, 0.5f * s3};
        float best = kInf;
        for (int e = 0; e < 3; ++e) {
            const float vx = px - ax[e], vy = py - ay[e];
            const float h = fminf(fmaxf((vx * ex[e] + vy * ey[e]) / (ex[e] * ex[e] + ey[e] * ey[e]), 0.f), 1.f);
            const float qx = vx - ex[e] * h, qy = vy - ey[e] * h;
            best = fminf(best, qx * qx + qy * qy);
        }
        return sqrtf(best) / (s3 / 6.f);
    }
    const F3 n = ld3(S.normals + 3 * (int64_t) iv[0]) * w + ld3(S.normals + 3 * (int64_t) iv[1]) * u + ld3(S.normals + 3 * (int64_t) iv[2]) * v;
    const float dp = -dot(n, ray_d);                                       // zero at silhouette points
    return dp * dp;
}

// Which warp of which path: the n-th reparameterize_ray call of path `widx` under the reparameterisation's own seed
// (common.py:1196-1203: 0xffffffff ^ seed).  Every auxiliary ray draws from its OWN stream, keyed by (seed, path, call,
// ray), so that the rays of a warp can be traced by different lanes (the device's second stage) or one after the other
// (the host build) with the same numbers.
struct WarpId { uint32_t key, widx; int n; };
// An auxiliary ray in two halves, so that the device can walk the rays of a wave TOGETHER between them (epsm_trace_packet.h):
// aux_begin draws the ray, aux_finish turns its closest hit into the warp's weight and its derivative.
struct AuxDraw { Ray ray; F3 tangent; float sy_; };                        // tangent = fs * omega.x + ft * omega.y
EPSM_HD AuxDraw aux_begin(const ReparamCfg &cfg, const WarpId &id, int r, F3 o, F3 d, F3 fs, F3 ft) {
    // antithetic pairs (reparam.py:82-84, 189-196): rays 2m and 2m + 1 share one sample; the even one mirrors it about the
    // ray (omega_local.x, .y negated), which cancels the odd part of the warp field's estimator
    const bool anti = (cfg.flags & EPSM_REPARAM_ANTITHETIC) != 0;
    const int rs = anti ? (r & ~1) : r;
    Pcg32 rng = seed_sampler(id.key ^ (0x9E3779B9u * (uint32_t) (id.n * 64 + rs + 1)), id.widx);
    const float kappa = cfg.kappa;
    const float sx = rng.next_1d(), sy_ = rng.next_1d();
    // warp.h:559-566 square_to_von_mises_fisher (1 - cos^2 formed without the cancellation of fp32)
    const float sy = fmaxf(1.f - sy_, 1e-6f);
    const float e = logf(sy + (1.f - sy) * expf(-2.f * kappa)) / kappa;             // cos_theta - 1 <= 0
    const float cos_theta = 1.f + e, sin_theta = safe_sqrt(-e * (2.f + e));
    const float phi = 2.f * kPi * sx;
    const float mirror = anti && !(r & 1) ? -1.f : 1.f;
    const F3 ol = f3(mirror * cosf(phi) * sin_theta, mirror * sinf(phi) * sin_theta, cos_theta);
    AuxDraw D;
    D.tangent = fs * ol.x + ft * ol.y;
    D.ray.o = o; D.ray.d = D.tangent + d * ol.z; D.ray.maxt = kInf;
    D.sy_ = sy_;
    return D;
}
EPSM_HD Aux aux_finish(const EpsmScene &S, const ReparamCfg &cfg, const AuxDraw &D, const TriHit &th, F3 o, F3 d) {
    const float kappa = cfg.kappa;
    Aux A;
    float B = 1.f;                                                         // reparam.py:104
    A.tri = kNoIndex; A.b1 = A.b2 = 0.f; A.inv_dist = 0.f; A.v = d;
    if (th.hit) {
        const uint32_t *iv = S.tri + 3 * (int64_t) th.tri;
        const F3 p = ld3(S.positions + 3 * (int64_t) iv[0]) * (1.f - th.u - th.v) + ld3(S.positions + 3 * (int64_t) iv[1]) * th.u +
                     ld3(S.positions + 3 * (int64_t) iv[2]) * th.v;
        const F3 r3 = p - o;
        const float dist = sqrtf(dot(r3, r3));
        if (dist > 0.f) {
            A.tri = th.tri; A.b1 = th.u; A.b2 = th.v; A.inv_dist = 1.f / dist; A.v = r3 * A.inv_dist;
            B = boundary_test(S, th, D.ray.d);
        }
    }
    const float inv_vmf = 1.f / (D.sy_ * expf(-2.f * kappa) + (1.f - D.sy_));              // reparam.py:111
    const float w_denom = inv_vmf - 1.f + B;
    const float w_denom_rcp = w_denom > 1e-4f ? 1.f / w_denom : 0.f;
    const float w = powf(w_denom_rcp, cfg.exponent) * inv_vmf;
    const float tmp1 = fminf(fmaxf(inv_vmf * w * w_denom_rcp * kappa * cfg.exponent, -1e10f), 1e10f);
    A.w = w;
    A.dw = D.tangent * tmp1;
    return A;
}
EPSM_HD Aux aux_ray(const EpsmScene &S, const ReparamCfg &cfg, const WarpId &id, int r, F3 o, F3 d, F3 fs, F3 ft, const BvhStack &st) {
    const AuxDraw D = aux_begin(cfg, id, r, o, d, fs, ft);
    return aux_finish(S, cfg, D, intersect<false>(S, D.ray, st), o, d);
}
// One call of reparameterize_ray's sampling loops (reparam.py:249-267 and 300-328 trace the same rays: kept instead of
// traced twice).
EPSM_HD void warp_collect(const EpsmScene &S, const ReparamCfg &cfg, const WarpId &id, F3 o, F3 d, const BvhStack &st, Warp &W) {
    W.n = cfg.rays; W.Z = 0.f; W.dZ = zero3<float>(); W.o = o; W.d = d;
    F3 fs, ft;
    coordinate_system(d, fs, ft);                                          // Frame3f(ray.d)
    for (int it = 0; it < cfg.rays; ++it) {
        W.a[it] = aux_ray(S, cfg, id, it, o, d, fs, ft, st);
        W.Z += W.a[it].w; W.dZ = W.dZ + W.a[it].dw;
    }
}

// ---------------------------------------------------------------------------
// accumulation into the gradient buffers
// ---------------------------------------------------------------------------
EPSM_HD void acc_add(float *p, float v) {
    // (a degenerate contribution -- a grazing ray's 1 / 0, a zero-length normal -- is dropped here instead of poisoning the
    // row for good; the reference scrubs NaN from the final parameter gradients, EPSM/optim.py:143-154)
    if (v == 0.f || !(fabsf(v) <= 3.0e38f)) return;
#if defined(__HIP_DEVICE_COMPILE__)
    atomicAdd(p, v);
#else
#pragma omp atomic
    *p += v;
#endif
}
struct GradOut { float *pos, *nrm; };                                      // (V,3) each, accumulated; nrm may be null
EPSM_HD void add_vertex(float *buf, uint32_t row, F3 g) { acc_add(buf + 3 * (int64_t) row, g.x); acc_add(buf + 3 * (int64_t) row + 1, g.y); acc_add(buf + 3 * (int64_t) row + 2, g.z); }
// d loss / d (a point glued to triangle `tri` at barycentrics b1, b2) = g  ->  the triangle's three vertices
EPSM_HD void add_follow_point(const EpsmScene &S, const GradOut &G, uint32_t tri, float b1, float b2, F3 g) {
    const EpsmMesh m = S.meshes[S.tri_mesh[tri]];
    if (!(m.flags & EPSM_MESH_POS_ATTACHED)) return;
    const uint32_t *iv = S.tri + 3 * (int64_t) tri;
    add_vertex(G.pos, iv[0], g * (1.f - b1 - b2)); add_vertex(G.pos, iv[1], g * b1); add_vertex(G.pos, iv[2], g * b2);
}

// Adjoint of one reparameterize_ray call (reparam.py:269-333) given d loss / d direction and d loss / d divergence:
// vertices of the triangles the auxiliary rays hit receive theirs; returns d loss / d ray.o and d loss / d ray.d.
EPSM_HD void warp_backward(const EpsmScene &S, const GradOut &G, const Warp &W, F3 g_dir, float g_div, F3 &g_o, F3 &g_d) {
    g_o = zero3<float>(); g_d = zero3<float>();
    if (W.n <= 0) return;
    const float Z = fmaxf(W.Z, 1e-8f), iZ = 1.f / Z;
    // direction = normalize(ray.d + V / Z), divergence = (div_V_1 - dot(V / Z, dZ)) / Z  at V = 0, div_V_1 = 0
    const F3 g_V = (g_dir - W.d * dot(W.d, g_dir)) * iZ - W.dZ * (g_div * iZ * iZ);
    const float g_div1 = g_div * iZ;
    // The divergence's two halves (dw_i g_div1 and the -dZ part of g_V) are each ~sqrt(kappa) times their sum and cancel
    // only over the auxiliary rays: the rays of a warp mostly hit the same one or two triangles, so their vertex
    // contributions are summed here in double, per triangle, before they go to the buffers.
    uint32_t tri[2] = {kNoIndex, kNoIndex};
    double acc[2][9], go[3] = {0.0, 0.0, 0.0};
    auto flush = [&](int e) {
        if (tri[e] == kNoIndex) return;
        const EpsmMesh m = S.meshes[S.tri_mesh[tri[e]]];
        if (m.flags & EPSM_MESH_POS_ATTACHED) {
            const uint32_t *iv = S.tri + 3 * (int64_t) tri[e];
            for (int j = 0; j < 3; ++j) add_vertex(G.pos, iv[j], f3((float) acc[e][3 * j], (float) acc[e][3 * j + 1], (float) acc[e][3 * j + 2]));
        }
        tri[e] = kNoIndex;
    };
    int victim = 0;
    for (int it = 0; it < W.n; ++it) {
        const Aux &A = W.a[it];
        const F3 g_v = g_V * A.w + A.dw * g_div1;                          // V_i = w V_direct, div_lhs_i = dot(d_w_omega, V_direct)
        if (A.tri == kNoIndex) { g_d = g_d + g_v; continue; }              // V_direct = ray.d (reparam.py:100)
        const F3 g_p = (g_v - A.v * dot(A.v, g_v)) * A.inv_dist;           // normalize(si.p - ray.o)
        int e = A.tri == tri[0] ? 0 : (A.tri == tri[1] ? 1 : -1);
        if (e < 0) {
            e = tri[0] == kNoIndex ? 0 : (tri[1] == kNoIndex ? 1 : victim);
            if (tri[e] != kNoIndex) { flush(e); victim ^= 1; }
            tri[e] = A.tri;
            for (int k = 0; k < 9; ++k) acc[e][k] = 0.0;
        }
        const double b[3] = {(double) (1.f - A.b1 - A.b2), (double) A.b1, (double) A.b2};
        for (int j = 0; j < 3; ++j) { acc[e][3 * j] += b[j] * g_p.x; acc[e][3 * j + 1] += b[j] * g_p.y; acc[e][3 * j + 2] += b[j] * g_p.z; }
        go[0] -= g_p.x; go[1] -= g_p.y; go[2] -= g_p.z;
    }
    flush(0); flush(1);
    g_o = f3((float) go[0], (float) go[1], (float) go[2]);
}

// ---------------------------------------------------------------------------
// what the differential step of a vertex needs to know about a vertex (captured from path_bounce, detached)
// ---------------------------------------------------------------------------
struct Vertex {
    bool valid;              // the path was alive and its ray hit something
    bool escaped;            // the path was alive and its ray left the scene (an environment emitter may shine along it)
    Ray ray;                 // the ray that arrived here
    TriHit th;
    SurfHit si;
    EpsmBsdf bsdf;
    uint32_t flags;
    F3 beta;                 // throughput on arrival
    F3 Le, Lr_dir;           // what this vertex added to L (already weighted by beta and MIS)
    F3 L_in, L_after;        // remaining radiance before / after those two terms (prb_reparam.py:435-436)
    EmitterSample es; bool active_em; float mis_em;
    BsdfSample bs; bool bs_valid;
    F3 wo_world;             // direction of the ray to the next vertex
    int depth;               // valid vertices before this one
};
EPSM_HD void vertex_clear(Vertex &v) { v.valid = false; v.escaped = false; v.L_in = v.L_after = v.Le = v.Lr_dir = zero3<float>(); v.active_em = false; v.bs_valid = false; v.depth = 0; }

// path_bounce observer: copies what the bounce computed
struct Capture {
    Vertex *out;
    EPSM_HD void vertex(const SurfHit &si, const EpsmBsdf &bsdf, uint32_t flags, F3 Le, F3 Lr_dir, const EmitterSample &es,
                        bool active_em, float mis_em, const BsdfSample &bs, bool path_active) {
        Vertex &v = *out;
        v.valid = path_active && si.valid;
        v.si = si; v.bsdf = bsdf; v.flags = flags;
        v.Le = path_active ? Le : zero3<float>(); v.Lr_dir = path_active ? Lr_dir : zero3<float>();
        v.es = es; v.active_em = active_em && path_active; v.mis_em = mis_em;
        v.bs = bs; v.bs_valid = bs.valid;
        v.wo_world = to_world(si, bs.wo);
    }
};

// Per-slot seeds of one dual evaluation: which input carries partial k (all others are constants)
struct Seeds { int d, dem, P[3], Nn[3]; };                                 // first partial index of each 3-vector, -1 = constant
template <class T> EPSM_HD V3<T> seed3(F3 a, int first) {
    V3<T> r = lift3<T>(a);
    if constexpr (!std::is_same<T, float>::value) {
        if (first >= 0) { r.x.d[first] = 1.f; r.y.d[first + 1] = 1.f; r.z.d[first + 2] = 1.f; }
    }
    return r;
}
template <class T> EPSM_HD T dot3c(F3 a, V3<T> b) { return b.x * T(a.x) + b.y * T(a.y) + b.z * T(a.z); }
template <class T> EPSM_HD V3<T> ratio3(F3 num, V3<T> f, F3 den, float floor_) {
    // num * f / max(floor, den) per channel (prb_reparam.py:541-542); zero where the detached value is zero
    return mk3<T>(den.x != 0.f ? f.x * T(num.x / fmaxf(floor_, den.x)) : T(0.f), den.y != 0.f ? f.y * T(num.y / fmaxf(floor_, den.y)) : T(0.f),
                  den.z != 0.f ? f.z * T(num.z / fmaxf(floor_, den.z)) : T(0.f));
}

// delta_L . [ (Le + Lr_dir + Lr_ind) + extra ] of vertex `cur` as a function of the seeded inputs (prb_reparam.py:362-572;
// the two determinants multiply terms whose values are known: handled by the caller).
template <class T>
EPSM_HD T eval_lo(const EpsmScene &S, const Vertex *prev, const Vertex &cur, const Vertex *next, F3 dL, const Seeds &sd) {
    const SurfHit &c = cur.si;
    const V3<T> o = lift3<T>(cur.ray.o);
    const V3<T> d = seed3<T>(cur.ray.d, sd.d);
    const V3<T> P0 = seed3<T>(c.p0, sd.P[0]), P1 = seed3<T>(c.p1, sd.P[1]), P2 = seed3<T>(c.p2, sd.P[2]);
    // the stored normals are post-flip (surface_interaction); surf_t flips after interpolating, as mesh.cpp does
    const float fl = (c.mesh_flags & EPSM_MESH_FLIP_NORMALS) ? -1.f : 1.f;
    const V3<T> N0 = seed3<T>(c.n0 * fl, sd.Nn[0]), N1 = seed3<T>(c.n1 * fl, sd.Nn[1]), N2 = seed3<T>(c.n2 * fl, sd.Nn[2]);
    const SurfT<T> h = surf_t<T>(o, d, P0, P1, P2, N0, N1, N2, c.mesh_flags, cur.th.u, cur.th.v, cur.th.t);
    T s = T(0.f);
    // ---- a `bitmap` reflectance follows the point the ray sees: rho(uv') / rho(uv) per channel multiplies both BSDF values below
    //      (bitmap.cpp:366-418: the texture lookup is attached to si.uv, mesh.cpp:736-745)
    V3<T> tex = mk3<T>(T(1.f), T(1.f), T(1.f));
    if (cur.bsdf.texture >= 0 && cur.bsdf.texture < S.n_textures && cur.bsdf.type == EPSM_BSDF_DIFFUSE_T) {
        float a0[2] = {0.f, 0.f}, a1[2] = {1.f, 0.f}, a2[2] = {0.f, 1.f};                    // no texture coordinates: uv = (b1, b2)
        if ((c.mesh_flags & EPSM_MESH_HAS_UV) && S.texcoords)
            for (int j = 0; j < 2; ++j) { a0[j] = S.texcoords[2 * (int64_t) c.vi[0] + j]; a1[j] = S.texcoords[2 * (int64_t) c.vi[1] + j]; a2[j] = S.texcoords[2 * (int64_t) c.vi[2] + j]; }
        const T b0 = T(1.f) - h.bu - h.bv;
        const T uu = b0 * T(a0[0]) + h.bu * T(a1[0]) + h.bv * T(a2[0]), vv = b0 * T(a0[1]) + h.bu * T(a1[1]) + h.bv * T(a2[1]);
        F3 du, dv;
        const F3 r0 = tex_eval(S.textures[cur.bsdf.texture], c.uvx, c.uvy, &du, &dv);
        const T eu = uu - T(c.uvx), ev = vv - T(c.uvy);                    // zero value, the derivative of uv
        if (r0.x > 0.f) tex.x = T(1.f) + (eu * T(du.x) + ev * T(dv.x)) * T(1.f / r0.x);
        if (r0.y > 0.f) tex.y = T(1.f) + (eu * T(du.y) + ev * T(dv.y)) * T(1.f / r0.y);
        if (r0.z > 0.f) tex.z = T(1.f) + (eu * T(du.z) + ev * T(dv.z)) * T(1.f / r0.z);
    }
    // ---- Lr_dir = beta * mis * bsdf(wi, to_local(d_em')) * em_weight (prb_reparam.py:413-418)
    if (cur.active_em && (cur.Lr_dir.x != 0.f || cur.Lr_dir.y != 0.f || cur.Lr_dir.z != 0.f)) {
        const V3<T> wo = to_local_t(h, seed3<T>(cur.es.d, sd.dem));
        const V3<T> f0_ = bsdf_eval_t<T>(cur.bsdf, h.wi, wo);
        const V3<T> f = mk3<T>(f0_.x * tex.x, f0_.y * tex.y, f0_.z * tex.z);
        const F3 k = mul3(mul3(cur.beta, cur.es.weight), dL) * cur.mis_em;
        T em = T(1.f);
        const bool env_sample = cur.es.emitter >= 0 && !cur.es.delta && cur.es.tri == kNoIndex && has_environment(S) && cur.es.emitter == S.env.emitter;
        if (env_sample && sd.dem >= 0) {
            // envmap.cpp:438-443 eval_direction at the reparameterised direction: L(d_em') / L(d_em) per channel, folded into
            // one factor weighted by what each channel contributes (k . f): d/d d_em of sum_c k_c f_c L_c(d') / L_c
            F3 g[3];
            const F3 L0 = env_eval_grad(S, cur.es.d, g);
            const V3<T> dd = seed3<T>(cur.es.d, sd.dem);
            const float Lc[3] = {L0.x, L0.y, L0.z};
            const float kc[3] = {k.x, k.y, k.z};
            const T fc[3] = {f.x, f.y, f.z};
            T acc = T(0.f);
            for (int c2 = 0; c2 < 3; ++c2) {
                const T ratio = Lc[c2] > 0.f ? T(1.f - dot(g[c2], cur.es.d) / Lc[c2]) + dot3c(g[c2] * (1.f / Lc[c2]), dd) : T(1.f);
                acc = acc + fc[c2] * T(kc[c2]) * ratio;
            }
            s = s + acc;
        } else
        if (cur.es.delta) {
            // point.cpp:154-164 eval_direction: intensity / |ds.p - it.p|^2 with it = si_cur ATTACHED (an area light's
            // value has no such dependence, area.cpp:182-192; its falloff sits in the detached pdf and the warp's divergence)
            const V3<T> r = lift3<T>(cur.es.p) - h.p;
            em = T(cur.es.dist * cur.es.dist) / dot(r, r);
            s = s + dot3c(k, f) * em;
        } else {
            s = s + dot3c(k, f) * em;
        }
    }
    // ---- Lr_ind = L * bsdf(wi, to_local(ray_next.d)) / detached (prb_reparam.py:554-568)
    if (cur.bs_valid) {
        const V3<T> wo = to_local_t(h, lift3<T>(cur.wo_world));
        const V3<T> f1_ = bsdf_eval_t<T>(cur.bsdf, h.wi, wo);
        const V3<T> f = mk3<T>(f1_.x * tex.x, f1_.y * tex.y, f1_.z * tex.z);
        const F3 den = cur.bs.weight * cur.bs.pdf;                         // bsdf_weight * bsdf_sample.pdf
        const V3<T> r = mk3<T>(den.x != 0.f ? f.x * T(cur.L_after.x / den.x) : T(0.f), den.y != 0.f ? f.y * T(cur.L_after.y / den.y) : T(0.f),
                               den.z != 0.f ? f.z * T(cur.L_after.z / den.z) : T(0.f));
        s = s + dot3c(dL, r);
    }
    // ---- extra: the neighbours' BSDFs as the point moves along the reparameterised ray over the DETACHED triangle
    //      (prb_reparam.py:515-542; the emission at the next vertex has no directional derivative for area lights)
    const bool want_prev = prev && prev->valid, want_next = next && next->valid;
    if (sd.d >= 0 && (want_prev || want_next)) {
        const V3<T> e1 = lift3<T>(c.p1 - c.p0), e2 = lift3<T>(c.p2 - c.p0);
        const V3<T> pvec = cross(d, e2);
        const V3<T> qvec = lift3<T>(cross(cur.ray.o - c.p0, c.p1 - c.p0));
        const T t = replace_value(dot(e2, qvec) / dot(e1, pvec), cur.th.t);
        const V3<T> p = o + d * t;
        if (want_prev) {
            const SurfHit &q = prev->si;
            const V3<T> wo_w = t_normalize(p - lift3<T>(q.p));
            const V3<T> wo = mk3<T>(dot3c(q.fs, wo_w), dot3c(q.ft, wo_w), dot3c(q.shn, wo_w));
            const V3<T> f = bsdf_eval_t<T>(prev->bsdf, lift3<T>(q.wi), wo);
            F3 f0; float pdf0; bsdf_eval_pdf(prev->bsdf, q.wi, f3(val(wo.x), val(wo.y), val(wo.z)), f0, pdf0);
            s = s + dot3c(dL, ratio3<T>(cur.L_in, f, f0, 1e-8f));
        }
        if (want_next) {
            const SurfHit &q = next->si;
            const V3<T> wi_w = t_normalize(p - lift3<T>(q.p));
            const V3<T> wi = mk3<T>(dot3c(q.fs, wi_w), dot3c(q.ft, wi_w), dot3c(q.shn, wi_w));
            const V3<T> f = bsdf_eval_t<T>(next->bsdf, wi, lift3<T>(next->bs.wo));
            F3 f0; float pdf0; bsdf_eval_pdf(next->bsdf, f3(val(wi.x), val(wi.y), val(wi.z)), next->bs.wo, f0, pdf0);
            s = s + dot3c(dL, ratio3<T>(next->L_after, f, f0, 1e-8f));
        }
    }
    return s;
}

// ---------------------------------------------------------------------------
// per-path arguments and the camera's Jacobian
// ---------------------------------------------------------------------------
struct ReparamArgs {
    TraceArgs A;                 // scene, sensor, seed, spp, max_depth, rr_depth, path range (nothing logged)
    ReparamCfg cfg;
    const float *radiance;       // (N,3) L of the primal pass under the same seed
    const float *adj_radiance;   // (N,3) d loss / d L
    const float *adj_film;       // (N,3) d loss / d film position (x, y) and d loss / d det of the primary ray
    GradOut G;
};

// d loss / d film position -> d loss / d reparameterised primary direction (common.py:405-418: the position is the
// projection of ray.o + d' by sensor.sample_direction; perspective.cpp's near plane is affine in the film position)
EPSM_HD F3 film_to_direction(const EpsmSensor &C, const PrimaryRay &pr, float gx, float gy) {
    const float *W = C.to_world;
    const F3 c0 = f3(W[0], W[4], W[8]), c1 = f3(W[1], W[5], W[9]), c2 = f3(W[2], W[6], W[10]);    // columns = camera axes in world space
    const F3 origin = f3(W[3], W[7], W[11]);
    const F3 r = (pr.ray.o - origin) + pr.ray.d;
    const F3 q = f3(dot(c0, r), dot(c1, r), dot(c2, r));                    // camera space (orthonormal to_world)
    const F3 base = xform_point(C.sample_to_camera, f3(0.f, 0.f, 0.f));
    const float s = base.z / q.z;
    // [u, v] = q.xy * s = base.xy + px * dx.xy + py * dy.xy   ->   h = A^-T g with A = [dx.xy dy.xy]
    const float a = C.dx[0], b = C.dy[0], c = C.dx[1], d = C.dy[1], idet = 1.f / (a * d - b * c);
    const float hx = (d * gx - c * gy) * idet, hy = (-b * gx + a * gy) * idet;
    const F3 gq = f3(s * hx, s * hy, -(q.x * hx + q.y * hy) * s / q.z);
    return c0 * gq.x + c1 * gq.y + c2 * gq.z;
}

// What happens to a warp once its ray and the adjoints of its direction / divergence are known.
//   o, d            the ray; g_dir, g_div: d loss / d direction, d loss / d divergence of its reparameterisation
//   ftri, fb1, fb2  the triangle + barycentrics its ORIGIN is glued to (kNoIndex: the camera)
//   em_inv_dist     emitter rays: 1 / |ds.p - o| -- their direction normalize(ds.p - o) follows the origin; else 0
// InlineSink: traced and back-propagated on the spot (the host build, and epsm_debug_*).
struct InlineSink {
    const EpsmScene &S; const ReparamCfg &cfg; const GradOut &G; const BvhStack &st; Warp &W; WarpId id;
    EPSM_HD void warp(F3 o, F3 d, F3 g_dir, float g_div, uint32_t ftri, float fb1, float fb2, float em_inv_dist) {
        warp_collect(S, cfg, id, o, d, st, W);
        id.n += 1;
        F3 g_o, g_d;
        warp_backward(S, G, W, g_dir, g_div, g_o, g_d);
        if (em_inv_dist != 0.f) g_o = g_o - (g_d - d * dot(d, g_d)) * em_inv_dist;
        if (ftri != kNoIndex) add_follow_point(S, G, ftri, fb1, fb2, g_o);
    }
};
// QueueSink: the request is written for the second stage (epsm_trace_reparam.hip), the n-th of path i at req[n * N + i].
struct alignas(16) WarpReq { float o[3], gdiv, d[3], em_inv_dist, gdir[3], fb1; uint32_t ftri; float fb2, pad0, pad1; };
constexpr int kMaxReq = 1 + 2 * 6;           // the camera ray + two warps per vertex (epsm.py:549: at most six vertices)
struct QueueSink {
    WarpReq *req; int64_t N, i; int n;
    EPSM_HD void warp(F3 o, F3 d, F3 g_dir, float g_div, uint32_t ftri, float fb1, float fb2, float em_inv_dist) {
        if (n >= kMaxReq) return;
        WarpReq q;
        q.o[0] = o.x; q.o[1] = o.y; q.o[2] = o.z; q.gdiv = g_div; q.d[0] = d.x; q.d[1] = d.y; q.d[2] = d.z; q.em_inv_dist = em_inv_dist;
        q.gdir[0] = g_dir.x; q.gdir[1] = g_dir.y; q.gdir[2] = g_dir.z; q.fb1 = fb1; q.ftri = ftri; q.fb2 = fb2; q.pad0 = q.pad1 = 0.f;
        req[(int64_t) n * N + i] = q;
        n += 1;
    }
};

// The differential step of one vertex (prb_reparam.py:341-589).  The camera ray's direction adjoint goes back to the
// caller (`primary_g_dir`): its warp also serves the film's reparameterisation.
template <class Sink>
EPSM_HD void differential(const ReparamArgs &R, Sink &sink, const Vertex *prev, const Vertex &cur, const Vertex *next, F3 dL, F3 *primary_g_dir) {
    const EpsmScene &S = R.A.S;
    const SurfHit &c = cur.si;
    const bool first = prev == nullptr;
    const bool has_normals = (c.mesh_flags & EPSM_MESH_VERTEX_NORMALS) != 0;
    const EpsmMesh m = S.meshes[c.mesh];
    // ---- dual evaluations: [d', d_em'], [P0, P1], [P2, N0], [N1, N2]
    float g[24];
    for (int k = 0; k < 24; ++k) g[k] = 0.f;
    const bool want_pos = (m.flags & EPSM_MESH_POS_ATTACHED) != 0, want_nrm = has_normals && (m.flags & EPSM_MESH_NRM_ATTACHED) && R.G.nrm;
    for (int chunk = 0; chunk < 4; ++chunk) {
        Seeds sd; sd.d = sd.dem = -1; sd.P[0] = sd.P[1] = sd.P[2] = sd.Nn[0] = sd.Nn[1] = sd.Nn[2] = -1;
        if (chunk == 0) { sd.d = 0; sd.dem = 3; }
        else if (chunk == 1) { if (!want_pos) continue; sd.P[0] = 0; sd.P[1] = 3; }
        else if (chunk == 2) { if (!want_pos && !want_nrm) continue; sd.P[2] = 0; sd.Nn[0] = 3; }
        else { if (!want_nrm) continue; sd.Nn[1] = 0; sd.Nn[2] = 3; }
        const Dual s = eval_lo<Dual>(S, prev, cur, next, dL, sd);
        for (int k = 0; k < kD; ++k) g[6 * chunk + k] = s.d[k];
    }
    if (want_pos) {
        add_vertex(R.G.pos, c.vi[0], f3(g[6], g[7], g[8])); add_vertex(R.G.pos, c.vi[1], f3(g[9], g[10], g[11]));
        add_vertex(R.G.pos, c.vi[2], f3(g[12], g[13], g[14]));
    }
    if (want_nrm) {
        add_vertex(R.G.nrm, c.vi[0], f3(g[15], g[16], g[17])); add_vertex(R.G.nrm, c.vi[1], f3(g[18], g[19], g[20]));
        add_vertex(R.G.nrm, c.vi[2], f3(g[21], g[22], g[23]));
    }
    // ---- the ray into this vertex (prb_reparam.py:341-358): d', det (det = 1 for the camera ray: the film carries it);
    //      its origin follows the previous shape (si_prev.spawn_ray: offset_p along the detached normal)
    const F3 g_dir = f3(g[0], g[1], g[2]);
    if (first) *primary_g_dir = g_dir;
    else if (cur.depth < R.cfg.max_depth)
        sink.warp(cur.ray.o, cur.ray.d, g_dir, dot(dL, cur.L_in) /* (Le + Lr_dir + Lr_ind) has the value L_in */, prev->th.tri, prev->th.u, prev->th.v, 0.f);
    // ---- the emitter ray (prb_reparam.py:394-411): em_ray.d = normalize(ds.p - o) with o glued to this triangle
    //      (si_cur_follow.spawn_ray_to)
    if (cur.active_em && cur.depth + 1 < R.cfg.max_depth && (cur.Lr_dir.x != 0.f || cur.Lr_dir.y != 0.f || cur.Lr_dir.z != 0.f)) {
        float dist;
        const Ray er = spawn_ray_to(c, cur.es.p, dist);
        sink.warp(er.o, er.d, f3(g[3], g[4], g[5]), dot(dL, cur.Lr_dir), cur.th.tri, cur.th.u, cur.th.v, 1.f / dist);
    }
}

// One path of RBIntegrator.render_backward's second pass (common.py:944-955).
template <class Sink>
EPSM_HD void reparam_one_path(const ReparamArgs &R, int64_t i, const BvhStack &st, Sink &sink) {
    const TraceArgs &A = R.A;
    const int64_t widx = A.path_offset + i;
    PathState s = path_begin(A, i, false);
    const F3 dL = ld3(R.adj_radiance + 3 * i);
    // ---- the camera ray: one warp serves sample_rays (film position + det, common.py:405-418) and the first vertex
    PrimaryRay pr;
    { Pcg32 r2 = seed_sampler(A.seed, (uint32_t) widx); pr = sample_primary_ray(A.C, widx, A.spp, r2); }
    const F3 g_film = film_to_direction(A.C, pr, R.adj_film[3 * i], R.adj_film[3 * i + 1]);
    const float g_det_film = R.adj_film[3 * i + 2];
    F3 g_first = zero3<float>();
    InlineVis vis{st};
    Vertex v[3];
    for (int k = 0; k < 3; ++k) vertex_clear(v[k]);
    F3 L_run = ld3(R.radiance + 3 * i);
    const int max_depth = path_max_depth(A);
    bool primary_done = false;
    // vertex j's step runs when vertex j+1 is known: slots (j-1, j, j+1) mod 3
    for (int j = 0; j <= max_depth; ++j) {
        Vertex &nx = v[j % 3];
        vertex_clear(nx);
        if (j < max_depth) {
            TriHit th; th.hit = false; th.tri = 0; th.t = kInf; th.u = th.v = 0.f;
            const bool was_active = s.active;
            if (s.active) th = intersect<false>(A.S, s.ray, st);
            nx.ray = s.ray; nx.th = th; nx.beta = s.beta; nx.depth = s.depth;
            Capture cap{&nx};
            path_bounce(A, i, j, s, th, vis, cap);
            nx.valid = nx.valid && was_active;
            nx.escaped = was_active && !th.hit;
            nx.L_in = L_run;
            L_run = L_run - nx.Le - nx.Lr_dir;
            nx.L_after = L_run;
        }
        if (j == 0) continue;
        const Vertex &cur = v[(j - 1) % 3];
        const Vertex *prev = j >= 2 ? &v[(j - 2) % 3] : nullptr;
        if (!cur.valid) {
            // a dead path: every term is zero.  A ray that left the scene towards an environment emitter carries that emitter's
            // radiance: its warp's divergence multiplies it and its direction moves the lookup (prb_reparam.py:341-358
            // reparameterises every ray before it is known to hit).  (The camera ray: below.)
            if (cur.escaped && prev && has_environment(A.S) && cur.depth < R.cfg.max_depth && (cur.Le.x != 0.f || cur.Le.y != 0.f || cur.Le.z != 0.f)) {
                F3 g[3];
                const F3 L0 = env_eval_grad(A.S, cur.ray.d, g);            // Le = beta mis L(d): d Le / d d = Le / L * grad L, per channel
                const F3 w = f3(L0.x > 0.f ? dL.x * cur.Le.x / L0.x : 0.f, L0.y > 0.f ? dL.y * cur.Le.y / L0.y : 0.f, L0.z > 0.f ? dL.z * cur.Le.z / L0.z : 0.f);
                sink.warp(cur.ray.o, cur.ray.d, g[0] * w.x + g[1] * w.y + g[2] * w.z, dot(dL, cur.L_in), prev->th.tri, prev->th.u, prev->th.v, 0.f);
            }
            continue;
        }
        differential(R, sink, prev, cur, &nx, dL, &g_first);
        if (j == 1 && R.cfg.max_depth > 0) {
            sink.warp(pr.ray.o, pr.ray.d, g_film + g_first, g_det_film, kNoIndex, 0.f, 0.f, 0.f);
            primary_done = true;
        }
    }
    if (!primary_done && R.cfg.max_depth > 0) {                            // the camera ray hit nothing: the film's term, and the
        F3 g_env = zero3<float>();                                         // environment's radiance along the reparameterised ray
        if (has_environment(A.S)) {
            F3 g[3];
            env_eval_grad(A.S, pr.ray.d, g);
            g_env = g[0] * dL.x + g[1] * dL.y + g[2] * dL.z;
        }
        sink.warp(pr.ray.o, pr.ray.d, g_film + g_env, g_det_film, kNoIndex, 0.f, 0.f, 0.f);
    }
}

}  // namespace rp
}  // namespace epsm
