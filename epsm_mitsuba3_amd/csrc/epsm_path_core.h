// epsm_path_core.h -- per-path arithmetic of the EPSM manifold-gradient hot path.
//
// One call = one light path: builds the 2x2 blocks of the projected half-vector
// constraint Jacobian in barycentric coordinates, solves the block system in
// ADJOINT form and emits the per-vertex parameter gradients.  It restates what
//     ManifoldIntegrator.calc_grad          (epsm.py:745-946)
//     ManifoldCausticIntegrator.calc_grad   (epsm.py:952-1200)
// compute, without ever forming the dense (2L x 2L) `constraint` matrix, the
// (N,2L,3) per-parameter Jacobians or a dense inverse (epsm.py:768-771,848):
//
//   * the reference contracts `dlduv . (-inv(cur) . J_p)` (epsm.py:850-851); with
//     y := cur^-T dlduv  this is  g_p = - sum_k y_k^T dC_k/dp, i.e. ONE reverse
//     sweep of constraint k seeded with the 2-vector y_k yields the gradient
//     w.r.t. every input of that constraint at once;
//   * "manifold": cur is block tridiagonal (2x2 blocks) -> block LU forward
//     recursion for the pivots, backward recursion for y; the sum over depths
//     id = 1..K and over the two sub-paths (light sample / continuing) is
//     folded into the seeds, so every constraint is swept once per version;
//   * "manifold_caustic": the row of the diffuse receiver is replaced by the
//     pseudo-constraint wo2 (epsm.py:1028,1051-1066,1116,1141-1157), which makes
//     cur a block upper-triangular band after a cyclic row shift -> one forward
//     recursion, no backward pass.
//
// The file is plain C++ templated on the scalar type; hipcc compiles it into
// the gfx950 kernels (epsm_kernels.hip) and tests/host_harness compiles the
// SAME code for the CPU in fp32/fp64 so the algebra can be checked against the
// oracle without a GPU.  It is not a CPU fallback: the product library only
// exports the HIP path.
#pragma once

#include <stdint.h>
#include <math.h>
#include <type_traits>

#if defined(__HIPCC__)
#define EPSM_HD __host__ __device__ __forceinline__
#else
#define EPSM_HD inline __attribute__((always_inline))
#endif

#define EPSM_LAMBDA __attribute__((always_inline))

namespace epsm {

// Compile-time loops over the vertex index.  `#pragma unroll` is only a request: with
// the fused output policy hipcc left the backward loop rolled, which turned the
// per-vertex state arrays into scratch memory (720 B/lane, 10^8 extra HBM writes per
// launch).  Recursion over an integral_constant cannot be left rolled.
template <int I, int N, typename F> EPSM_HD void static_for_up(F &&f) {
    if constexpr (I <= N) { f(std::integral_constant<int, I>{}); static_for_up<I + 1, N>(f); }
}
template <int I, typename F> EPSM_HD void static_for_down(F &&f) {
    if constexpr (I >= 1) { f(std::integral_constant<int, I>{}); static_for_down<I - 1>(f); }
}

constexpr int kMaxVertices = 5;          // EPSM_MAX_VERTICES
constexpr uint32_t kBsdfNull = 0x1u;     // bsdf.h:40
constexpr uint32_t kBsdfDiffuse = 0x6u;  // bsdf.h:101

// ----------------------------------------------------------------------------
// tiny linear algebra
// ----------------------------------------------------------------------------
template <typename R> struct V3 { R x, y, z; };
template <typename R> struct V2 { R x, y; };
template <typename R> struct M2 { R a, b, c, d; };   // [[a,b],[c,d]]

template <typename R> EPSM_HD V3<R> mk3(R x, R y, R z) { V3<R> v; v.x = x; v.y = y; v.z = z; return v; }
template <typename R> EPSM_HD V3<R> zero3() { return mk3<R>(R(0), R(0), R(0)); }
template <typename R> EPSM_HD V3<R> operator+(V3<R> a, V3<R> b) { return mk3<R>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <typename R> EPSM_HD V3<R> operator-(V3<R> a, V3<R> b) { return mk3<R>(a.x - b.x, a.y - b.y, a.z - b.z); }
template <typename R> EPSM_HD V3<R> operator-(V3<R> a) { return mk3<R>(-a.x, -a.y, -a.z); }
template <typename R> EPSM_HD V3<R> operator*(V3<R> a, R s) { return mk3<R>(a.x * s, a.y * s, a.z * s); }
template <typename R> EPSM_HD R dot(V3<R> a, V3<R> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename R> EPSM_HD V3<R> cross(V3<R> a, V3<R> b) {
    return mk3<R>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// a + b*s
template <typename R> EPSM_HD V3<R> madd(V3<R> a, V3<R> b, R s) { return mk3<R>(a.x + b.x * s, a.y + b.y * s, a.z + b.z * s); }

template <typename R> EPSM_HD V2<R> mk2(R x, R y) { V2<R> v; v.x = x; v.y = y; return v; }
template <typename R> EPSM_HD V2<R> operator+(V2<R> a, V2<R> b) { return mk2<R>(a.x + b.x, a.y + b.y); }
template <typename R> EPSM_HD V2<R> operator-(V2<R> a, V2<R> b) { return mk2<R>(a.x - b.x, a.y - b.y); }
template <typename R> EPSM_HD V2<R> operator*(V2<R> a, R s) { return mk2<R>(a.x * s, a.y * s); }
// row vector times matrix
template <typename R> EPSM_HD V2<R> vmul(V2<R> v, M2<R> m) { return mk2<R>(v.x * m.a + v.y * m.c, v.x * m.b + v.y * m.d); }
template <typename R> EPSM_HD M2<R> mmul(M2<R> p, M2<R> q) {
    M2<R> r;
    r.a = p.a * q.a + p.b * q.c; r.b = p.a * q.b + p.b * q.d;
    r.c = p.c * q.a + p.d * q.c; r.d = p.c * q.b + p.d * q.d;
    return r;
}
template <typename R> EPSM_HD M2<R> msub(M2<R> p, M2<R> q) { M2<R> r; r.a = p.a - q.a; r.b = p.b - q.b; r.c = p.c - q.c; r.d = p.d - q.d; return r; }
EPSM_HD float rcp_(float x);
EPSM_HD double rcp_(double x);
template <typename R> EPSM_HD M2<R> minv(M2<R> m) {
    R det = m.a * m.d - m.b * m.c;
    R id = rcp_(det);
    M2<R> r; r.a = m.d * id; r.b = -m.b * id; r.c = -m.c * id; r.d = m.a * id;
    return r;
}

EPSM_HD bool finite_(float x) { return fabsf(x) <= 3.402823466e+38f; }
EPSM_HD bool finite_(double x) { return fabs(x) <= 1.7976931348623157e+308; }
template <typename R> EPSM_HD bool finite2(V2<R> v) { return finite_(v.x) && finite_(v.y); }
// 1/sqrt(x) and 1/x.  On gfx950 the IEEE expansions of sqrtf and '/' cost ~10
// VALU instructions each; v_rsq_f32 / v_rcp_f32 (1 ulp) plus one Newton step is
// within an ulp of them in 4 / 3 instructions.  x = 0 still yields a non-finite
// value (inf -> NaN through the Newton step), which is what the NaN -> 0 rule of
// the reference (epsm.py:746-748, 856) relies on.
#if defined(__HIP_DEVICE_COMPILE__)
EPSM_HD float rsqrt_(float x) {
    float y = __builtin_amdgcn_rsqf(x);
    float e = fmaf(-x * y, y, 1.0f);
    return fmaf(0.5f * y, e, y);
}
EPSM_HD float rcp_(float x) {
    float y = __builtin_amdgcn_rcpf(x);
    return fmaf(fmaf(-x, y, 1.0f), y, y);
}
#else
EPSM_HD float rsqrt_(float x) { return 1.0f / sqrtf(x); }
EPSM_HD float rcp_(float x) { return 1.0f / x; }
#endif
EPSM_HD double rsqrt_(double x) { return 1.0 / sqrt(x); }
#if defined(__HIP_DEVICE_COMPILE__) && defined(EPSM_FAST_RCP64)
// (the IEEE expansion of 1.0 / x in float64 is 13 instructions -- two scalings, v_rcp_f64, two Newton steps, a fused fix-up;
// the 2x2 recursions of the backward kernel take v_rcp_f64 and the two Newton steps alone: determinants there are far from the
// ends of the exponent range, and x = 0 still gives inf -> NaN)
EPSM_HD double rcp_(double x) {
    double y = __builtin_amdgcn_rcp(x);
    y = __builtin_fma(__builtin_fma(-x, y, 1.0), y, y);
    return __builtin_fma(__builtin_fma(-x, y, 1.0), y, y);
}
#else
EPSM_HD double rcp_(double x) { return 1.0 / x; }
#endif
EPSM_HD float realmax_(float) { return 3.402823466e+38f; }
EPSM_HD double realmax_(double) { return 1.7976931348623157e+308; }

// ----------------------------------------------------------------------------
// kernel arguments (typed twin of EpsmVertexRecord, include/epsm.h)
// ----------------------------------------------------------------------------
template <typename R> struct VertexPtrs {
    const R *p0, *p1, *p2, *n0, *n1, *n2, *b0, *b1, *eta, *light;
    const uint32_t *bsdf;
    const uint8_t *active, *active_em, *ismesh;
};

template <int K> struct Flags;
template <typename R> struct GradArgs {
    int64_t N;
    const R *cam;
    VertexPtrs<R> v[kMaxVertices];
    const R *dlduv;          // row n at dlduv + n*dlduv_stride
    int64_t dlduv_stride;
    const R *dldp;           // (N,3)
    R clip;                  // <= 0: clamp disabled
    R *out_param;            // (P,N,3)
    R *out_light;            // (K,N,3)
    R *out_diffuse;          // (K,N,3)
    EPSM_HD const VertexPtrs<R> &vtx(int k) const { return v[k]; }
    template <int K> EPSM_HD Flags<K> flags(int64_t i) const;      // from the record arrays (defined below load_flags)
    // the record of vertex k (1-based) of path i: see Raw / soa_raw below
    EPSM_HD V3<R> cam_at(int64_t i) const;
    EPSM_HD auto raw(int k, int64_t i) const;
    EPSM_HD auto geo(int k, int64_t i) const;
    EPSM_HD auto nrm(int k, int64_t i, R b0, R b1) const;
    EPSM_HD R eta(int k, int64_t i) const;
    // the tangents of the first vertex (dldp) and of vertex k (columns 2(k-1), 2(k-1)+1 of dlduv)
    EPSM_HD V3<R> dldp_at(int64_t i) const { const R *p = dldp + 3 * i; return mk3<R>(p[0], p[1], p[2]); }
    template <bool FULL_D> EPSM_HD V2<R> d_at(int64_t i, int k, int dcols) const;      // defined below load_d
};

// Record arrays are GLOBAL memory: gl() says so.  A pointer that reaches the code through memory (the fused kernel
// keeps its 85 array pointers in an LDS table) is a generic pointer to the compiler, and a load through it is a FLAT
// load: it counts on vmcnt AND lgkmcnt, so every wait for an LDS operation also waits for the record loads in flight
// and the other way round -- the prefetch of the next vertex and the LDS queue / table traffic serialise each other.
#if defined(__HIP_DEVICE_COMPILE__)
template <typename T> EPSM_HD const __attribute__((address_space(1))) T *gl(const T *p) { return (const __attribute__((address_space(1))) T *) p; }
#else
template <typename T> EPSM_HD const T *gl(const T *p) { return p; }
#endif
// One element of a record array.  (Marking these streams non-temporal, so that the 10 GB of records of a launch do not
// push the scene's triangle table and the parameter rows out of L2, measured 0 to +4 %: not kept.)
template <typename T> EPSM_HD T lds_(const T *p, int64_t i) { return gl(p)[i]; }
template <typename R> EPSM_HD V3<R> load3(const R *base, int64_t i) {
    return mk3<R>(lds_(base, 3 * i), lds_(base, 3 * i + 1), lds_(base, 3 * i + 2));
}

// torch.nan_to_num followed by the +-clip outlier removal (epsm.py:856, 932-944):
// NaN -> 0; +-inf -> +-max, which the clamp then zeroes; |g| > clip -> 0.  One
// ordered compare does all three (it is false for NaN).  `clip` is the largest
// finite value when the caller disabled the clamp, so non-finite values still
// come out as 0 there (nan_to_num would give +-max for an infinity).
template <typename R> EPSM_HD R finalize(R g, R clip) {
    R a = g < R(0) ? -g : g;
    return a <= clip ? g : R(0);
}
template <typename R> EPSM_HD void store3(R *base, int64_t slot, int64_t N, int64_t i, V3<R> g, R clip) {
    R *p = base + (slot * N + i) * 3;
    p[0] = finalize(g.x, clip);
    p[1] = finalize(g.y, clip);
    p[2] = finalize(g.z, clip);
}

// Output policy of the per-path functions.  DenseOut writes the reference's own
// result arrays (calc_grad's three lists, epsm.py:946); the fused kernel plugs in a
// policy that accumulates straight into the parameter-gradient buffers instead
// (epsm_grad_scatter.hip), so the 84 B/vertex of dense results never touch HBM.
//   pre_id(k, live)
//        called once per vertex right after the flags are known: the 4-byte triangle id of
//        the vertex's addressing, loaded well ahead of the table lookup that depends on it;
//   pre_emit(k, live)
//        the emitter-sample record of vertex k, requested ONE STEP before pre_aux(k, ...) so that the lookup
//        that depends on it (emitter triangle -> vertex rows) can be issued at the top of step k;
//   pre_tri(k, live, id) / pre_aux(k, live, emit)
//        called at the TOP of the step that will emit vertex k, so that whatever the
//        policy must fetch for it (parameter addressing) is in flight while the step's
//        sweeps run -- a load issued where the gradient becomes known would be a
//        dependent HBM round trip per vertex in a kernel with two waves per SIMD;
//   vertex(k, has_nm, Gx, gn, gm, glight, ctx, tri, aux)
//        d/d p_j of vertex k = b_j * Gx  (slots 5(k-1)+0..2), d/d n (slot +3), d/d m
//        (slot +4; both only when has_nm), light_grad[k-1];
//   diffuse(idx, g, b0, b1, tri)   diffuse_grad[idx] = position gradient of vertex idx+1,
//        whose barycentrics / addressing are passed along;
//   diffuse_first(g, id)  diffuse_grad[0] (known before anything else is computed; id of vertex 1);
//   undo_needed(poisoned, P)   caustic only: a live term turned out non-finite after some of its
//                     rows were already emitted -> all parameter rows are zero.  Dense output
//                     zeroes them and returns false; an accumulating policy returns true when
//                     any lane of the wave is poisoned and the path function then emits the
//                     negated rows in a second turn.
// Every lane of a wave calls the policy at the same program points (lanes without a
// gradient pass zeros), so a policy may use wave-wide operations.
template <typename R> struct VCtx { R b0, b1; V3<R> n, e1, e2; };   // retained geometry of the vertex

template <typename R> struct DenseOut {
    struct Id {};
    struct Tri {};
    struct Aux {};
    struct Emit {};
    const GradArgs<R> &A;
    int64_t i;
    // any(p): may the step be skipped when p is false for this lane?  Dense output never
    // skips (zeros must be stored); the fused policy skips when p is false on the whole wave.
    EPSM_HD bool any(bool) const { return true; }
    EPSM_HD Id pre_id(int, bool) const { return Id{}; }
    EPSM_HD Tri pre_tri(int, bool, Id) const { return Tri{}; }
    EPSM_HD Emit pre_emit(int, bool) const { return Emit{}; }
    EPSM_HD Aux pre_aux(int, bool, Emit) const { return Aux{}; }
    EPSM_HD void vertex(int k, bool has_nm, V3<R> Gx, V3<R> gn, V3<R> gm, V3<R> glight,
                        const VCtx<R> &c, const Tri &, const Aux &) const {
        store3(A.out_param, 5 * (k - 1) + 0, A.N, i, Gx * c.b0, A.clip);
        store3(A.out_param, 5 * (k - 1) + 1, A.N, i, Gx * c.b1, A.clip);
        store3(A.out_param, 5 * (k - 1) + 2, A.N, i, Gx * (R(1) - c.b0 - c.b1), A.clip);
        if (has_nm) {
            store3(A.out_param, 5 * (k - 1) + 3, A.N, i, gn, A.clip);
            store3(A.out_param, 5 * (k - 1) + 4, A.N, i, gm, A.clip);
        }
        store3(A.out_light, k - 1, A.N, i, glight, A.clip);
    }
    EPSM_HD void diffuse(int idx, V3<R> g, R, R, const Tri &) const { store3(A.out_diffuse, idx, A.N, i, g, A.clip); }
    EPSM_HD void diffuse_first(V3<R> g, Id) const { store3(A.out_diffuse, 0, A.N, i, g, A.clip); }
    EPSM_HD bool undo_needed(bool poisoned, int P) const {
        if (poisoned)
            for (int q = 0; q < P; ++q) store3(A.out_param, q, A.N, i, zero3<R>(), A.clip);
        return false;
    }
};

// ----------------------------------------------------------------------------
// geometry of one logged vertex
// ----------------------------------------------------------------------------
template <typename R> struct Geo {      // from "points" + "uv": x = p0 b0 + p1 b1 + p2 (1-b0-b1)  (epsm.py:758-759)
    V3<R> x, e1, e2;                    // e_j = dx/db_j = p_j - p2
    R b0, b1;
};
template <typename R> EPSM_HD Geo<R> load_geo(const VertexPtrs<R> &v, int64_t i) {
    Geo<R> g;
    V3<R> p0 = load3(v.p0, i), p1 = load3(v.p1, i), p2 = load3(v.p2, i);
    g.b0 = lds_(v.b0, i);
    g.b1 = lds_(v.b1, i);
    R b2 = R(1) - g.b0 - g.b1;
    g.x = p0 * g.b0 + p1 * g.b1 + p2 * b2;
    g.e1 = p0 - p2;
    g.e2 = p1 - p2;
    return g;
}
template <typename R> struct Nrm { V3<R> n, dn1, dn2; };   // epsm.py:761-762 (un-normalised interpolation)
template <typename R> EPSM_HD Nrm<R> load_nrm(const VertexPtrs<R> &v, int64_t i, R b0, R b1) {
    Nrm<R> o;
    V3<R> n0 = load3(v.n0, i), n1 = load3(v.n1, i), n2 = load3(v.n2, i);
    R b2 = R(1) - b0 - b1;
    o.n = n0 * b0 + n1 * b1 + n2 * b2;
    o.dn1 = n0 - n2;
    o.dn2 = n1 - n2;
    return o;
}

// Everything one logged vertex contributes to the solve.  The path functions ask their argument object for it
// (A.raw / A.geo / A.nrm / A.eta / A.cam_at), so that the same code runs on the reference's tensor layout (nine
// (N,3) arrays per vertex: soa_* below) and on the packed per-vertex records of the native pipeline
// (epsm_grad_scatter.hip, PackedArgs).
template <typename R> struct Raw { Geo<R> g; Nrm<R> nr; R eta; V3<R> light; };
template <typename R, typename Args> EPSM_HD Raw<R> soa_raw(const Args &A, int kk, int64_t i) {
    Raw<R> r;
    r.g = load_geo(A.vtx(kk - 1), i);
    r.nr = load_nrm(A.vtx(kk - 1), i, r.g.b0, r.g.b1);
    r.eta = lds_(A.vtx(kk - 1).eta, i);
    r.light = load3(A.vtx(kk - 1).light, i);
    return r;
}

// local frame of epsm.py:746-756: rows t, n^ x t, n^;  t = normalize(0,-n^_z,n^_y)
template <typename R> struct Frame { V3<R> nn, t, bt; R inv_n, inv_v; };
template <typename R> EPSM_HD Frame<R> make_frame(V3<R> n) {
    Frame<R> f;
    f.inv_n = rsqrt_(dot(n, n));
    f.nn = n * f.inv_n;
    V3<R> v = mk3<R>(R(0), -f.nn.z, f.nn.y);
    f.inv_v = rsqrt_(dot(v, v));
    f.t = v * f.inv_v;
    f.bt = cross(f.nn, f.t);
    return f;
}
// adjoint of the frame rows -> adjoint of the un-normalised normal
template <typename R> EPSM_HD V3<R> frame_rev(const Frame<R> &f, V3<R> tb, V3<R> btb, V3<R> nb) {
    nb = nb + cross(f.t, btb);
    tb = tb + cross(btb, f.nn);
    V3<R> vb = (tb - f.t * dot(f.t, tb)) * f.inv_v;
    nb.z -= vb.y;
    nb.y += vb.z;
    return (nb - f.nn * dot(f.nn, nb)) * f.inv_n;
}

// C = [ normalize(R wi + eta R wo) ]_xy, wi = normalize(xp-xc), wo = normalize(xn-xc)  (epsm.py:809-821)
template <typename R> struct HalfVec {
    V3<R> wi, wo, u;          // u = wi + eta wo (world space); |R u| = |u|
    R inv_a, inv_b, inv_u, eta;
    R rx, ry, rz;             // normalised half vector in the local frame
};
template <typename R> EPSM_HD HalfVec<R> halfvec_fwd(V3<R> xp, V3<R> xc, V3<R> xn, const Frame<R> &f, R eta) {
    HalfVec<R> h;
    V3<R> a = xp - xc, b = xn - xc;
    h.inv_a = rsqrt_(dot(a, a));
    h.inv_b = rsqrt_(dot(b, b));
    h.wi = a * h.inv_a;
    h.wo = b * h.inv_b;
    h.eta = eta;
    h.u = madd(h.wi, h.wo, eta);
    h.inv_u = rsqrt_(dot(h.u, h.u));
    h.rx = dot(f.t, h.u) * h.inv_u;
    h.ry = dot(f.bt, h.u) * h.inv_u;
    h.rz = dot(f.nn, h.u) * h.inv_u;
    return h;
}
template <typename R> struct Sweep { V3<R> gxp, gxc, gxn, gn; };

// reverse sweep of s0*C_x + s1*C_y
template <typename R> EPSM_HD Sweep<R> halfvec_rev(const Frame<R> &f, const HalfVec<R> &h, R s0, R s1) {
    Sweep<R> o;
    R re = s0 * h.rx + s1 * h.ry;
    R r0 = (s0 - h.rx * re) * h.inv_u, r1 = (s1 - h.ry * re) * h.inv_u, r2 = (-h.rz * re) * h.inv_u;
    V3<R> wib = f.t * r0 + f.bt * r1 + f.nn * r2;
    V3<R> ab = (wib - h.wi * dot(h.wi, wib)) * h.inv_a;
    V3<R> wob = wib * h.eta;
    V3<R> bb = (wob - h.wo * dot(h.wo, wob)) * h.inv_b;
    o.gxp = ab;
    o.gxn = bb;
    o.gxc = -(ab + bb);
    o.gn = frame_rev(f, h.u * r0, h.u * r1, h.u * r2);
    return o;
}
// reverse sweep of s0*wo2_x + s1*wo2_y, wo2 = R normalize(xn-xc)   (epsm.py:1028,1116)
template <typename R> EPSM_HD Sweep<R> wo2_rev(const Frame<R> &f, const HalfVec<R> &h, R s0, R s1) {
    Sweep<R> o;
    V3<R> wob = f.t * s0 + f.bt * s1;
    V3<R> bb = (wob - h.wo * dot(h.wo, wob)) * h.inv_b;
    o.gxp = zero3<R>();
    o.gxn = bb;
    o.gxc = -bb;
    o.gn = frame_rev(f, h.wo * s0, h.wo * s1, zero3<R>());
    return o;
}
template <typename R> EPSM_HD Sweep<R> zero_sweep() {
    Sweep<R> o; o.gxp = o.gxc = o.gxn = o.gn = zero3<R>(); return o;
}
template <typename R> EPSM_HD Sweep<R> comb(const Sweep<R> &a, R s0, const Sweep<R> &b, R s1) {
    Sweep<R> o;
    o.gxp = a.gxp * s0 + b.gxp * s1; o.gxc = a.gxc * s0 + b.gxc * s1;
    o.gxn = a.gxn * s0 + b.gxn * s1; o.gn = a.gn * s0 + b.gn * s1;
    return o;
}

// row i of a 2x2 block = (g . e1, g . e2)
template <typename R> EPSM_HD M2<R> block2(V3<R> g0, V3<R> g1, V3<R> e1, V3<R> e2) {
    M2<R> m; m.a = dot(g0, e1); m.b = dot(g0, e2); m.c = dot(g1, e1); m.d = dot(g1, e2); return m;
}
template <typename R> EPSM_HD M2<R> madd2(M2<R> p, M2<R> q) { M2<R> r; r.a = p.a + q.a; r.b = p.b + q.b; r.c = p.c + q.c; r.d = p.d + q.d; return r; }

template <typename R, bool FULL_D, typename Args> EPSM_HD V2<R> load_d(const Args &A, int64_t i, int k /*1-based*/, int dcols) {
    if (!FULL_D && k > 1) return mk2<R>(R(0), R(0));
    const auto *row = gl(A.dlduv) + i * A.dlduv_stride;
    int c = 2 * (k - 1);
    R x = c < dcols ? row[c] : R(0);
    R y = c + 1 < dcols ? row[c + 1] : R(0);
    return mk2<R>(x, y);
}
template <typename R> template <bool FULL_D> EPSM_HD V2<R> GradArgs<R>::d_at(int64_t i, int k, int dcols) const {
    return load_d<R, FULL_D>(*this, i, k, dcols);
}

// ----------------------------------------------------------------------------
// per-depth masks from the flag words (no geometry needed)
// ----------------------------------------------------------------------------
template <int K> struct Flags {
    bool diffuse[K + 2], null_[K + 2], active[K + 2], active_em[K + 2], mesh[K + 2];
};
// five bits per vertex in one word (the fused kernel parks a window's flags in LDS)
template <int K> EPSM_HD uint32_t pack_flags(const Flags<K> &f) {
    uint32_t w = 0;
#pragma unroll
    for (int k = 1; k <= K; ++k)
        w |= ((f.diffuse[k] ? 1u : 0u) | (f.null_[k] ? 2u : 0u) | (f.active[k] ? 4u : 0u) | (f.active_em[k] ? 8u : 0u) |
              (f.mesh[k] ? 16u : 0u)) << (5 * (k - 1));
    return w;
}
template <typename R, int K, typename Args> EPSM_HD Flags<K> load_flags(const Args &A, int64_t i) {
    Flags<K> f;
#pragma unroll
    for (int k = 1; k <= K; ++k) {
        uint32_t b = lds_(A.vtx(k - 1).bsdf, i);
        f.diffuse[k] = (b & kBsdfDiffuse) != 0;
        f.null_[k] = (b & kBsdfNull) != 0;
        f.active[k] = lds_(A.vtx(k - 1).active, i) != 0;
        f.active_em[k] = lds_(A.vtx(k - 1).active_em, i) != 0;
        f.mesh[k] = lds_(A.vtx(k - 1).ismesh, i) != 0;
    }
    f.diffuse[0] = f.null_[0] = f.active[0] = f.active_em[0] = f.mesh[0] = false;
    f.diffuse[K + 1] = f.null_[K + 1] = f.active[K + 1] = f.active_em[K + 1] = f.mesh[K + 1] = false;
    return f;
}
template <int K> EPSM_HD Flags<K> unpack_flags(uint32_t w) {
    Flags<K> f;
    f.diffuse[0] = f.null_[0] = f.active[0] = f.active_em[0] = f.mesh[0] = false;
    f.diffuse[K + 1] = f.null_[K + 1] = f.active[K + 1] = f.active_em[K + 1] = f.mesh[K + 1] = false;
#pragma unroll
    for (int k = 1; k <= K; ++k) {
        const uint32_t b = w >> (5 * (k - 1));
        f.diffuse[k] = (b & 1u) != 0; f.null_[k] = (b & 2u) != 0; f.active[k] = (b & 4u) != 0;
        f.active_em[k] = (b & 8u) != 0; f.mesh[k] = (b & 16u) != 0;
    }
    return f;
}
template <typename R> template <int K> EPSM_HD Flags<K> GradArgs<R>::flags(int64_t i) const { return load_flags<R, K>(*this, i); }
template <typename R> EPSM_HD V3<R> GradArgs<R>::cam_at(int64_t i) const { return load3(cam, i); }
template <typename R> EPSM_HD auto GradArgs<R>::raw(int k, int64_t i) const { return soa_raw<R>(*this, k, i); }
template <typename R> EPSM_HD auto GradArgs<R>::geo(int k, int64_t i) const { return load_geo(v[k - 1], i); }
template <typename R> EPSM_HD auto GradArgs<R>::nrm(int k, int64_t i, R b0, R b1) const { return load_nrm(v[k - 1], i, b0, b1); }
template <typename R> EPSM_HD R GradArgs<R>::eta(int k, int64_t i) const { return lds_(v[k - 1].eta, i); }

// Scalar type the 2x2 block recursion of `manifold` is COMPUTED in (pivots, forward vectors, adjoint seeds; they are
// kept between the steps in the working precision R).  The reference inverts `cur` with a pivoted LU
// (torch.linalg.inv, epsm.py:848,912); the block recursion cannot pivot -- the systems of the depths are the nested
// leading blocks of ONE matrix, which is what makes a single O(K) recursion serve all of them -- and in float32 its
// rounding errors grow like cond^2 where the pivoted LU's grow like cond: 0.7 % of the `specular` paths with
// cond_2 < 1e4 were off by more than 4 eps cond, up to O(1).  Doing the ~60 flops per vertex of the recursion in
// float64 (the 2x2 blocks themselves come out of float32 sweeps) removes that: 0.05 %, worst 4e-2 at cond 6e3
// (tests/test_kernel_core_host.py, test_gpu_parity.py); +0.6 % kernel time fused, +4 % for the dense kernel.
#ifdef EPSM_REC_FLOAT      // (A/B build: the recursion in the scalar type itself -- 0.7 % of the `specular` paths leave the bound)
template <typename R> struct RecType { typedef R type; };
#else
template <typename R> struct RecType { typedef double type; };
#endif
template <typename Q, typename R> EPSM_HD M2<Q> cvm(M2<R> m) { M2<Q> o; o.a = Q(m.a); o.b = Q(m.b); o.c = Q(m.c); o.d = Q(m.d); return o; }
template <typename Q, typename R> EPSM_HD V2<Q> cvv(V2<R> v) { return mk2<Q>(Q(v.x), Q(v.y)); }

// ============================================================================
// "manifold"  (epsm.py:745-946)
// ============================================================================
// Number of vertices whose geometry a path needs (`nv` of manifold_path / caustic_path): the steps a lane is live
// in.  Only used to GROUP paths of similar length into the same wave (epsm_grad_scatter.hip); the path functions
// derive their own masks, so a mismatch here could only cost time, never change a result.
template <int K> EPSM_HD int manifold_extent(const Flags<K> &fl) {
    bool valid = true;
    int hasdiffuse = 0, nv = 0;
#pragma unroll
    for (int id = 1; id <= K; ++id) {
        valid = valid && fl.mesh[id];
        hasdiffuse += fl.diffuse[id] ? 1 : 0;
        valid = valid && (hasdiffuse < 2);
        const bool spec = valid && (hasdiffuse == 0);
        if (spec && fl.active[id] && fl.active_em[id]) nv = id;
        if ((id < K) && spec && fl.active[id + 1] && fl.diffuse[id + 1]) nv = id + 1;
    }
    return nv;
}
template <int K> EPSM_HD int caustic_extent(const Flags<K> &fl) {
    bool valid = true;
    int hasdiffuse = 0, nv = 0;
#pragma unroll
    for (int id = 1; id <= K; ++id) {
        valid = valid && fl.mesh[id];
        hasdiffuse += fl.diffuse[id] ? 1 : 0;
        valid = valid && (hasdiffuse < 2);
        if ((id < K) && fl.diffuse[1] && valid && fl.active[id + 1] && (fl.diffuse[id + 1] || fl.null_[id + 1])) nv = id + 1;
    }
    return nv;
}

template <typename R, int K, bool FULL_D, typename Out, typename Args>
EPSM_HD void manifold_path(const Args &A, int64_t i, int dcols, const Out &out) {
    const Flags<K> fl = A.template flags<K>(i);

    // term masks (epsm.py:793-802, 852-855, 916-920).  wN[id]: light-sampling
    // sub-path at depth id; wC[id]: continuing sub-path x_{id-1},x_id,x_{id+1}.
    bool wN[K + 1], wC[K + 1];
    int nv = 0;                       // last vertex whose geometry is needed
    bool nvN = false;                 // ... and whether it has a light-sampling term (wN[nv])
    {
        bool valid = true;
        int hasdiffuse = 0;
#pragma unroll
        for (int id = 1; id <= K; ++id) {
            valid = valid && fl.mesh[id];
            hasdiffuse += fl.diffuse[id] ? 1 : 0;
            valid = valid && (hasdiffuse < 2);
            const bool spec = valid && (hasdiffuse == 0);
            wN[id] = spec && fl.active[id] && fl.active_em[id];
            wC[id] = (id < K) && spec && fl.active[id + 1] && fl.diffuse[id + 1];
            if (wN[id]) { nv = id; nvN = true; }
            if (wC[id]) { nv = id + 1; nvN = false; }
        }
    }

    // triangle ids of the vertices that can receive rows (an accumulating policy looks their vertices up later)
    typename Out::Id tid[K + 2];
    static_for_up<1, K>([&](auto kc) EPSM_LAMBDA { constexpr int k = decltype(kc)::value; tid[k] = out.pre_id(k, k <= nv || k == 1); });
    tid[0] = tid[K + 1] = out.pre_id(1, false);
    // the emitter-sample record of the vertex pass 2 starts with (the others are requested one step ahead there)
    const typename Out::Emit em_first = out.pre_emit(nv >= 1 ? nv : 1, nvN);

    // diffuse_grad[0] = dldp where the first hit is diffuse (epsm.py:791-792)
    out.diffuse_first(fl.diffuse[1] ? A.dldp_at(i) : zero3<R>(), tid[1]);

    // ---- pass 1: forward recursion (pivots of the continuing rows, z vectors)
    struct Keep { V3<R> x, e1, e2, n, light; R eta, b0, b1; };
    Keep kp[K + 2];
    typedef typename RecType<R>::type Q;
    typedef R St;                     // kept between the steps / passes in the working precision
    M2<St> Sinv[K + 1];
    V2<St> z[K + 1], zN[K + 1];
    z[0] = mk2<St>(St(0), St(0));

    // Everything a vertex contributes to pass 1, fetched ONE STEP AHEAD of its use: with two
    // waves per SIMD a load consumed in the step that issued it is a fully exposed HBM
    // round trip (the kernel was latency-bound at 35 % VALU utilisation before this).
    const V3<R> cam = A.cam_at(i);
    Raw<R> rnext;
    if (nv >= 1) rnext = A.raw(1, i);
    M2<Q> Aup;                        // A^C_{k-1,k}: continuing row k-1, column block k
    Aup.a = Aup.b = Aup.c = Aup.d = Q(0);

    static_for_up<1, K>([&](auto kc) EPSM_LAMBDA {
        constexpr int k = decltype(kc)::value;
        if (k <= nv) {
            const Raw<R> r = rnext;
            const Geo<R> g = r.g;
            const Nrm<R> nr = r.nr;
            kp[k].x = g.x; kp[k].e1 = g.e1; kp[k].e2 = g.e2; kp[k].b0 = g.b0; kp[k].b1 = g.b1;
            const bool has_next = (k < K) && (k + 1 <= nv);
            if (has_next) rnext = A.raw(k < K ? k + 1 : K, i);
            kp[k].n = nr.n;
            kp[k].eta = r.eta;
            kp[k].light = r.light;
            const Frame<R> fr = make_frame(nr.n);
            const V3<R> xp = (k == 1) ? cam : kp[k - 1].x;
            const V2<Q> dk = cvv<Q>(A.template d_at<FULL_D>(i, k, dcols));

            V2<Q> rhs = dk;
            M2<Q> T;                  // Sinv_{k-1} A^C_{k-1,k}
            T.a = T.b = T.c = T.d = Q(0);
            if (k > 1) {
                rhs = dk - vmul(cvv<Q>(z[k - 1]), Aup);
                T = mmul(cvm<Q>(Sinv[k - 1]), Aup);
            }
            // light-sampling version (next point = emitter sample); its rows never serve
            // deeper terms (those see the continuing rows), so it is needed only when live
            zN[k] = mk2<St>(St(0), St(0));
            if (wN[k]) {
                const HalfVec<R> h = halfvec_fwd(xp, g.x, kp[k].light, fr, kp[k].eta);
                const Sweep<R> s0 = halfvec_rev(fr, h, R(1), R(0));
                const Sweep<R> s1 = halfvec_rev(fr, h, R(0), R(1));
                M2<Q> Akk = cvm<Q>(madd2(block2(s0.gxc, s1.gxc, g.e1, g.e2), block2(s0.gn, s1.gn, nr.dn1, nr.dn2)));
                M2<Q> S = Akk;
                if (k > 1) S = msub(Akk, mmul(cvm<Q>(block2(s0.gxp, s1.gxp, kp[k - 1].e1, kp[k - 1].e2)), T));
                zN[k] = cvv<St>(vmul(rhs, minv(S)));
            }
            // continuing version (next point = x_{k+1})
            if (has_next) {
                const Geo<R> &gn_ = rnext.g;
                const HalfVec<R> h = halfvec_fwd(xp, g.x, gn_.x, fr, kp[k].eta);
                const Sweep<R> s0 = halfvec_rev(fr, h, R(1), R(0));
                const Sweep<R> s1 = halfvec_rev(fr, h, R(0), R(1));
                M2<Q> Akk = cvm<Q>(madd2(block2(s0.gxc, s1.gxc, g.e1, g.e2), block2(s0.gn, s1.gn, nr.dn1, nr.dn2)));
                M2<Q> S = Akk;
                if (k > 1) S = msub(Akk, mmul(cvm<Q>(block2(s0.gxp, s1.gxp, kp[k - 1].e1, kp[k - 1].e2)), T));
                const M2<Q> Si = minv(S);
                Sinv[k] = cvm<St>(Si);
                z[k] = cvv<St>(vmul(rhs, Si));
                Aup = cvm<Q>(block2(s0.gxn, s1.gxn, gn_.e1, gn_.e2));
            }
        }
    });

    // ---- pass 2: backward recursion of the adjoint seeds + seeded sweeps
    // (Tried: keeping the two unit-seed sweeps of the LAST vertex's light-sampling constraint in registers across the
    // pass boundary -- pass 2 starts at that vertex and its seeded sweep there is their linear combination, ~200 of the
    // ~700 VALU instructions of the step.  24 more live floats: fused kernel 3.95 -> 4.46 ms, dense 2.87 -> 3.09 ms.)
    V2<Q> carry = mk2<Q>(Q(0), Q(0));  // sum over deeper terms of their y_k
    int W = 0;                         // number of live terms with depth > k
    V3<R> GP = zero3<R>();             // d/dx_k through constraint k+1 (x_k as previous vertex)
    typename Out::Tri tri_next = out.pre_tri(K, false, tid[0]);   // addressing of vertex k+1 (for diffuse_grad[k])
    typename Out::Emit em_next = out.pre_emit(1, false);
    R nb0 = R(0), nb1 = R(0);
    static_for_down<K>([&](auto kc) EPSM_LAMBDA {
        constexpr int k = decltype(kc)::value;
        const typename Out::Tri tri = out.pre_tri(k, k <= nv, tid[k]);
        const typename Out::Aux aux = out.pre_aux(k, k <= nv, k == nv ? em_first : em_next);
        if (k > 1) em_next = out.pre_emit(k - 1, k <= nv && wN[k - 1]);
        V3<R> Gxk = zero3<R>(), gnrm = Gxk, gm = Gxk, glight = Gxk, gdiff = Gxk;
        VCtx<R> ctx; ctx.b0 = ctx.b1 = R(0); ctx.n = ctx.e1 = ctx.e2 = zero3<R>();
        if (k <= nv) {
            const bool fN = wN[k] && finite2(zN[k]);
            const bool fC = wC[k] && finite2(z[k]);
            const V2<R> sN = fN ? cvv<R>(zN[k]) : mk2<R>(R(0), R(0));
            V2<Q> sCq = carry;
            if (fC) sCq = sCq + cvv<Q>(z[k]);
            const V2<R> sC = cvv<R>(sCq);
            const bool has_next = (k < K) && (k + 1 <= nv);
            const Frame<R> fr = make_frame(kp[k].n);
            const V3<R> xp = (k == 1) ? cam : kp[k - 1].x;
            Sweep<R> a = zero_sweep<R>(), c = zero_sweep<R>();
            if (sN.x != R(0) || sN.y != R(0)) {
                const HalfVec<R> h = halfvec_fwd(xp, kp[k].x, kp[k].light, fr, kp[k].eta);
                a = halfvec_rev(fr, h, sN.x, sN.y);
            }
            if (has_next && (sC.x != R(0) || sC.y != R(0))) {
                const HalfVec<R> h = halfvec_fwd(xp, kp[k].x, kp[k + 1].x, fr, kp[k].eta);
                c = halfvec_rev(fr, h, sC.x, sC.y);
            }
            Gxk = -(a.gxc + c.gxc + GP);
            ctx.b0 = kp[k].b0; ctx.b1 = kp[k].b1; ctx.n = kp[k].n; ctx.e1 = kp[k].e1; ctx.e2 = kp[k].e2;
            gnrm = -(a.gn + c.gn);
            if (has_next) gm = mk3<R>(sC.x, sC.y, R(0));   // dC/dm = -I on continuing rows (epsm.py:883)
            glight = -a.gxn;
            if (fC) gdiff = -c.gxn;                        // carry == 0 whenever fC (a diffuse x_{k+1} ends the chain)
            GP = a.gxp + c.gxp;
            W += (fN ? 1 : 0) + (fC ? 1 : 0);
            if (k > 1) {
                if (W > 0) {
                    V2<Q> q = mk2<Q>(Q(dot(GP, kp[k - 1].e1)), Q(dot(GP, kp[k - 1].e2)));
                    carry = cvv<Q>(z[k - 1]) * Q(W) - vmul(q, cvm<Q>(Sinv[k - 1]));
                } else {
                    carry = mk2<Q>(Q(0), Q(0));
                }
            }
        }
        if (out.any(k <= nv)) out.vertex(k, true, Gxk, gnrm, gm, glight, ctx, tri, aux);
        if (k < K && out.any(gdiff.x != R(0) || gdiff.y != R(0) || gdiff.z != R(0))) out.diffuse(k, gdiff, nb0, nb1, tri_next);
        tri_next = tri; nb0 = ctx.b0; nb1 = ctx.b1;
    });
}

// ============================================================================
// "manifold_caustic"  (epsm.py:952-1200)
// ============================================================================
// Outputs can only be non-zero when the first hit is diffuse (dlduv is zeroed
// otherwise, epsm.py:999) and then only through continuing sub-paths: every
// light-sampling term is masked by `hasdiffuse>0` (epsm.py:1083,1090), so
// light_grad == 0.  With the receiver x_1 diffuse, row block 1 of `cur` holds
// the pseudo-constraint wo2 of the CURRENT depth (column block id only) and
// rows 2..id the half-vector constraints: unknowns y_2.. follow a forward
// recursion that does not depend on the depth, y_1 closes it per depth.
template <typename R, int K, bool FULL_D, typename Out, typename Args>
EPSM_HD void caustic_path(const Args &A, int64_t i, int dcols, const Out &out) {
    const Flags<K> fl = A.template flags<K>(i);
    constexpr int P = 5 * K - 2;

    bool wP[K + 1], wD[K + 1];
    int nv = 0, idstar = 0;
    {
        bool valid = true;
        int hasdiffuse = 0;
#pragma unroll
        for (int id = 1; id <= K; ++id) {
            valid = valid && fl.mesh[id];
            hasdiffuse += fl.diffuse[id] ? 1 : 0;
            valid = valid && (hasdiffuse < 2);
            const bool base = (id < K) && fl.diffuse[1] && valid && fl.active[id + 1];
            wP[id] = base && fl.diffuse[id + 1];                        // epsm.py:1172-1174
            wD[id] = base && (fl.diffuse[id + 1] || fl.null_[id + 1]);  // epsm.py:1180-1182
            if (wD[id]) nv = id + 1;
            if (wP[id]) idstar = id;    // at most one: the next diffuse vertex invalidates deeper terms
        }
    }

    // A term whose solve turns out non-finite at depth id* contributes nothing to ANY parameter (the whole inverse is
    // NaN -> nan_to_num, epsm.py:1076-1079), but the rows of vertices 1..id*-2 leave the lane before that is known.
    // Dense output zeroes the path's rows afterwards (poison); an accumulating policy asks for a second turn of the
    // loop in which the lanes concerned re-derive exactly those rows and emit them NEGATED.  Wave-uniform and rare
    // (degenerate geometry only), so the common case runs the body once.
    typename Out::Id tid[K + 2];
    static_for_up<1, K>([&](auto kc) EPSM_LAMBDA { constexpr int k = decltype(kc)::value; tid[k] = out.pre_id(k, k <= nv || k == 1); });
    tid[0] = tid[K + 1] = out.pre_id(1, false);
    const int nv_all = nv, idstar_all = idstar;
    bool poisoned = false;             // a live term turned out non-finite: zero every param gradient (nan_to_num)
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
    const bool undo = pass == 1;
    if (undo) { nv = poisoned ? nv_all : 0; idstar = poisoned ? idstar_all : 0; }
    const int undo_last = idstar - 2;  // undo turn: vertices 1..id*-2 were emitted before the term was known to be bad

    if (!undo) out.diffuse_first(fl.diffuse[1] ? A.dldp_at(i) : zero3<R>(), tid[1]);   // epsm.py:998-1000

    const V3<R> cam = A.cam_at(i);
    Geo<R> gcur, gnext;
    if (nv >= 1) gnext = A.geo(1, i);
    typename Out::Tri tri_prev = out.pre_tri(1, false, tid[0]), tri_cur = out.pre_tri(1, nv >= 1, tid[1]);
    V3<R> n_prev = zero3<R>();
    V2<R> vprev = mk2<R>(R(0), R(0)), vcur = vprev;   // v_{k-1}, v_k   (v_1 = 0)
    V2<R> rprev = vprev;                               // r_{k-1}
    M2<R> Aup; Aup.a = Aup.b = Aup.c = Aup.d = R(0);   // A_{k-1,k}
    V3<R> xprev = cam, e1prev = zero3<R>(), e2prev = zero3<R>();
    R b0prev = R(0), b1prev = R(0);
    V3<R> Gx_prev = zero3<R>();        // -(d/dx_{k-1}) gathered so far for vertex k-1
    V3<R> gn_prev = zero3<R>(), gm_prev = zero3<R>();

    static_for_up<1, K>([&](auto kc) EPSM_LAMBDA {
        constexpr int k = decltype(kc)::value;
        V3<R> gnrm = zero3<R>(), gm = gnrm, gdiff = gnrm;
        V3<R> Gx = zero3<R>(), ncur = zero3<R>();
        R b0 = R(0), b1 = R(0);
        const bool live = (k < K) && (k + 1 <= nv);    // depth k has a continuing sub-path we need
        const typename Out::Tri tri_nxt = out.pre_tri(k < K ? k + 1 : K, live, tid[k < K ? k + 1 : K]);   // vertex k+1: diffuse_grad[k] now, rows later
        const typename Out::Aux aux_prev = out.pre_aux(k >= 2 ? k - 1 : 1, k >= 2 && (k - 1) <= idstar, out.pre_emit(1, false));   // (light_grad == 0)
        if (live) {
            gcur = gnext;
            gnext = A.geo(k < K ? k + 1 : K, i);
            b0 = gcur.b0; b1 = gcur.b1;
            const Nrm<R> nr = A.nrm(k, i, gcur.b0, gcur.b1);
            ncur = nr.n;
            const R eta = A.eta(k, i);
            const Frame<R> fr = make_frame(nr.n);
            const HalfVec<R> h = halfvec_fwd(xprev, gcur.x, gnext.x, fr, eta);
            const V2<R> dk = A.template d_at<FULL_D>(i, k, dcols);
            // pseudo-constraint rows (always needed: they close the system at depth k)
            const Sweep<R> w0 = wo2_rev(fr, h, R(1), R(0));
            const Sweep<R> w1 = wo2_rev(fr, h, R(0), R(1));
            const M2<R> Wk = madd2(block2(w0.gxc, w1.gxc, gcur.e1, gcur.e2), block2(w0.gn, w1.gn, nr.dn1, nr.dn2));
            V2<R> rk = dk;
            Sweep<R> c0 = zero_sweep<R>(), c1 = c0;
            if (k >= 2) {
                c0 = halfvec_rev(fr, h, R(1), R(0));
                c1 = halfvec_rev(fr, h, R(0), R(1));
                const M2<R> Akm = block2(c0.gxp, c1.gxp, e1prev, e2prev);
                const M2<R> Akk = madd2(block2(c0.gxc, c1.gxc, gcur.e1, gcur.e2), block2(c0.gn, c1.gn, nr.dn1, nr.dn2));
                vcur = vmul(rprev, minv(Akm));                       // y_k for every depth >= k
                rk = dk - vmul(vcur, Akk) - vmul(vprev, Aup);
                Aup = block2(c0.gxn, c1.gxn, gnext.e1, gnext.e2);
            }
            const V2<R> uk = vmul(rk, minv(Wk));                      // y_1 at depth k
            const bool fin = finite2(uk) && finite2(vcur);
            const bool inP = (k <= idstar);                           // rows of this vertex carry weight 1
            if (inP && !fin && k == idstar) poisoned = true;
            // continuing constraint of vertex k, seed v_k (k>=2), weight [k <= id*]
            const Sweep<R> cs = comb(c0, vcur.x, c1, vcur.y);
            // pseudo-constraint of depth k, seed u_k, weight [k == id*]
            const Sweep<R> ws = comb(w0, uk.x, w1, uk.y);
            if (inP) {
                Gx = -cs.gxc;
                gnrm = -cs.gn;
                if (k >= 2) gm = mk3<R>(vcur.x, vcur.y, R(0));
                // x_{k-1} as previous vertex of constraint k
                Gx_prev = Gx_prev - cs.gxp;
                if (k == idstar) {
                    Gx = Gx - ws.gxc;
                    gnrm = gnrm - ws.gn;
                }
            }
            if (wD[k] && fin) {
                // epsm.py:1139-1157: row block id (gxn, plus the stale wo2[0] gradient on its
                // second row) and the pseudo rows (d wo2 / d x_{k+1})
                if (k >= 2) {
                    const Sweep<R> wx = comb(w0, uk.x + vcur.y, w1, uk.y);
                    gdiff = -(wx.gxn + cs.gxn);
                } else {
                    gdiff = -ws.gxn;
                }
            }
            rprev = rk;
            vprev = vcur;
        }
        // vertex k-1 is complete once constraint k has been swept.  First turn: nothing more once the term is known
        // to be bad; undo turn: minus what the first turn emitted before it knew.
        const bool emit_prev = undo ? (k - 1) <= undo_last : ((k - 1) <= idstar && !poisoned);
        if (k >= 2 && out.any(emit_prev)) {
            const R m = emit_prev ? (undo ? R(-1) : R(1)) : R(0);
            VCtx<R> c; c.b0 = b0prev; c.b1 = b1prev; c.n = n_prev; c.e1 = e1prev; c.e2 = e2prev;
            out.vertex(k - 1, true, Gx_prev * m, gn_prev * m, gm_prev * m, zero3<R>(), c, tri_prev, aux_prev);
        }
        if (k < K && !undo && out.any(live)) out.diffuse(k, gdiff, live ? gnext.b0 : R(0), live ? gnext.b1 : R(0), tri_nxt);
        Gx_prev = Gx; gn_prev = gnrm; gm_prev = gm;
        tri_prev = tri_cur; tri_cur = tri_nxt;
        // geometry of vertex k, kept for its emission at step k+1 (flat-normal rows need e1,e2,n)
        if (live) {
            xprev = gcur.x; e1prev = gcur.e1; e2prev = gcur.e2;
        }
        n_prev = ncur;
        b0prev = b0; b1prev = b1;
    });
    // last vertex: only p0,p1,p2 are registered and no continuing row exists for it (id* <= K-1: never part of an undo)
    if (!undo) {
        VCtx<R> c; c.b0 = b0prev; c.b1 = b1prev; c.n = n_prev; c.e1 = e1prev; c.e2 = e2prev;
        out.vertex(K, false, poisoned ? zero3<R>() : Gx_prev, zero3<R>(), zero3<R>(), zero3<R>(), c, tri_prev, out.pre_aux(K, false, out.pre_emit(1, false)));
    }
    if (undo || !out.undo_needed(poisoned, P)) break;
    }
}

}  // namespace epsm
