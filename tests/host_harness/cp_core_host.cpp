// TEST HARNESS ONLY: compiles epsm_mitsuba3_amd/csrc/epsm_cp_core.h -- the constraint-parallel arithmetic the gfx950
// backward kernel runs, one lane per (path, constraint vertex) -- for the host CPU in fp32 and fp64.  The lanes of a path
// are array entries here and the lane-to-lane exchange of the recursions is an index shift; the functions that do the
// arithmetic are the kernel's own.  Output: calc_grad's dense lists, so the same golden / oracle tests as for
// epsm_path_core.h apply (tests/test_kernel_core_host.py).  Not shipped, not a fallback.
#include <string.h>

#include "../../epsm_mitsuba3_amd/csrc/epsm_cp_core.h"
#include "../../include/epsm.h"

using namespace epsm;

template <typename R> struct HostArgs {
    int64_t N;
    int K, P;
    const R *cam;
    VertexPtrs<R> v[kMaxVertices];
    const R *dlduv;
    int64_t stride;
    int dcols;
    const R *dldp;
    R clip;
    R *op, *ol, *od;
    const VertexPtrs<R> &vtx(int k) const { return v[k]; }      // what soa_raw asks its argument object for
};

template <typename R> static uint32_t flag_word(const HostArgs<R> &A, int64_t i) {
    uint32_t w = 0;
    for (int k = 1; k <= A.K; ++k) {
        const VertexPtrs<R> &v = A.v[k - 1];
        const uint32_t b = v.bsdf[i];
        w |= (((b & kBsdfDiffuse) ? 1u : 0u) | ((b & kBsdfNull) ? 2u : 0u) | (v.active[i] ? 4u : 0u) | (v.active_em[i] ? 8u : 0u) |
              (v.ismesh[i] ? 16u : 0u)) << (5 * (k - 1));
    }
    return w;
}
template <typename R> static cp::Own<R> own_of(const HostArgs<R> &A, int k, int64_t i) {
    const Raw<R> r = soa_raw<R>(A, k, i);
    cp::Own<R> o;
    o.x = r.g.x; o.e1 = r.g.e1; o.e2 = r.g.e2; o.b0 = r.g.b0; o.b1 = r.g.b1;
    o.n = r.nr.n; o.dn1 = r.nr.dn1; o.dn2 = r.nr.dn2; o.eta = r.eta; o.light = r.light;
    return o;
}
template <typename R> static cp::Nbr<R> nbr_of(const HostArgs<R> &A, int k, int64_t i) {
    cp::Nbr<R> n;
    if (k == 0) { n.x = load3(A.cam, i); n.e1 = n.e2 = zero3<R>(); return n; }
    const Geo<R> g = load_geo(A.v[k - 1], i);
    n.x = g.x; n.e1 = g.e1; n.e2 = g.e2;
    return n;
}
template <typename R> static V2<R> d_of(const HostArgs<R> &A, int k, int64_t i) {
    const R *row = A.dlduv + i * A.stride;
    const int c = 2 * (k - 1);
    return mk2<R>(c < A.dcols ? row[c] : R(0), c + 1 < A.dcols ? row[c + 1] : R(0));
}
template <typename R> static void put(R *base, int64_t slot, const HostArgs<R> &A, int64_t i, V3<R> g) { store3(base, slot, A.N, i, g, A.clip); }

template <typename R> static cp::Pts<R> pts_of(const cp::Own<R> &o, const cp::Nbr<R> &prev, const cp::Nbr<R> &next) {
    cp::Pts<R> p;
    p.x = o.x; p.n = o.n; p.light = o.light; p.eta = o.eta; p.xp = prev.x; p.xn = next.x;
    return p;
}

template <typename R> static void manifold_one(const HostArgs<R> &A, int64_t i) {
    const uint32_t plan = cp::manifold_plan(flag_word(A, i));
    const int nv = cp::plan_nv(plan), m = cp::plan_m(plan);
    if (cp::plan_diffuse1(plan)) put(A.od, 0, A, i, load3(A.dldp, i));
    cp::Own<R> own[kMaxVertices + 1];
    cp::Pts<R> pts[kMaxVertices + 1];
    cp::MBlocks<R> ev[kMaxVertices + 1];
    cp::MFwd<R> fw[kMaxVertices + 2];
    cp::MOut<R> out[kMaxVertices + 2];
    for (int k = 1; k <= m; ++k) {                        // pass 1: the blocks
        const bool has_next = k + 1 <= nv;
        own[k] = own_of(A, k, i);
        cp::Nbr<R> next; next.x = next.e1 = next.e2 = zero3<R>();
        if (has_next) next = nbr_of(A, k + 1, i);
        const cp::Nbr<R> prev = nbr_of(A, k - 1, i);
        ev[k] = cp::manifold_blocks(own[k], prev, next, cp::plan_a(plan, k), has_next);
        pts[k] = pts_of(own[k], prev, next);
    }
    fw[0] = cp::mfwd_zero<R>();
    for (int k = 1; k <= m; ++k)
        fw[k] = cp::manifold_fwd(ev[k], d_of(A, k, i), k == 1, fw[k - 1], k > 1 ? ev[k - 1].Aup : ev[k].Aup, cp::plan_a(plan, k), k + 1 <= nv);
    cp::MBwd<R> nb; nb.q = mk2<R>(R(0), R(0)); nb.W = 0;
    for (int k = m; k >= 1; --k) {                        // backward recursion on the blocks, then pass 2: the seeded sweeps
        cp::MBwd<R> mine;
        const cp::MSeeds<R> sd = cp::manifold_bwd(ev[k], fw[k], nb, cp::plan_a(plan, k), cp::plan_b(plan, k), k + 1 <= nv, mine);
        nb = mine;
        out[k] = cp::manifold_contract(pts[k], sd, k + 1 <= nv);
    }
    out[m + 1].GP = zero3<R>();
    for (int k = 1; k <= m; ++k) {
        const cp::MOut<R> &o = out[k];
        const V3<R> Gx = o.Gx - out[k + 1].GP;           // d/dx_k through constraint k+1 (the lane exchange of the kernel)
        const R b0 = own[k].b0, b1 = own[k].b1;
        put(A.op, 5 * (k - 1) + 0, A, i, Gx * b0);
        put(A.op, 5 * (k - 1) + 1, A, i, Gx * b1);
        put(A.op, 5 * (k - 1) + 2, A, i, Gx * (R(1) - b0 - b1));
        put(A.op, 5 * (k - 1) + 3, A, i, o.gn);
        put(A.op, 5 * (k - 1) + 4, A, i, o.gm);
        put(A.ol, k - 1, A, i, o.glight);
        if (k < A.K) put(A.od, k, A, i, o.gdiff);
    }
}

template <typename R> static void caustic_one(const HostArgs<R> &A, int64_t i) {
    const uint32_t plan = cp::caustic_plan(flag_word(A, i));
    const int m = cp::plan_m(plan), idstar = cp::plan_idstar(plan);
    if (cp::plan_diffuse1(plan)) put(A.od, 0, A, i, load3(A.dldp, i));
    cp::Own<R> own[kMaxVertices + 1];
    cp::Pts<R> pts[kMaxVertices + 1];
    cp::CBlocks<R> ev[kMaxVertices + 1];
    cp::CFwd<R> fw[kMaxVertices + 2];
    cp::COut<R> out[kMaxVertices + 2];
    for (int k = 1; k <= m; ++k) {
        own[k] = own_of(A, k, i);
        const cp::Nbr<R> prev = nbr_of(A, k - 1, i), next = nbr_of(A, k + 1, i);
        ev[k] = cp::caustic_blocks(own[k], prev, next, k == 1);
        pts[k] = pts_of(own[k], prev, next);
    }
    fw[0] = cp::cfwd_zero<R>();
    bool poisoned = false;
    for (int k = 1; k <= m; ++k) {
        fw[k] = cp::caustic_fwd(ev[k], d_of(A, k, i), k == 1, fw[k - 1], k > 1 ? ev[k - 1].Aup : ev[k].Aup);
        if (k == idstar && !fw[k].fin) poisoned = true;
        out[k] = cp::caustic_finish(pts[k], fw[k], k == 1, k <= idstar, k == idstar, cp::plan_b(plan, k));
    }
    out[m + 1].gxp_prev = zero3<R>();
    for (int k = 1; k <= m; ++k) {
        if (k < A.K) put(A.od, k, A, i, out[k].gdiff);
        if (poisoned || k > idstar) continue;            // a non-finite term at id* drops every parameter row (nan_to_num, epsm.py:1076-1079)
        const V3<R> Gx = k + 1 <= m ? out[k].Gx + out[k + 1].gxp_prev : out[k].Gx;
        const R b0 = own[k].b0, b1 = own[k].b1;
        put(A.op, 5 * (k - 1) + 0, A, i, Gx * b0);
        put(A.op, 5 * (k - 1) + 1, A, i, Gx * b1);
        put(A.op, 5 * (k - 1) + 2, A, i, Gx * (R(1) - b0 - b1));
        put(A.op, 5 * (k - 1) + 3, A, i, out[k].gn);
        put(A.op, 5 * (k - 1) + 4, A, i, out[k].gm);
    }
}

template <typename R>
static int run(int variant, int64_t N, int K, const void *cam, const EpsmVertexRecord *verts, const void *dlduv, int64_t stride, int dcols,
               const void *dldp, double clip, void *op, void *ol, void *od) {
    if (K < 1 || K > kMaxVertices) return EPSM_EINVAL;
    HostArgs<R> A{};
    A.N = N; A.K = K; A.P = variant == EPSM_VARIANT_MANIFOLD ? 5 * K : 5 * K - 2;      // epsm_num_param_grads
    A.cam = (const R *) cam;
    for (int k = 0; k < K; ++k) {
        const EpsmVertexRecord &v = verts[k];
        VertexPtrs<R> &o = A.v[k];
        o.p0 = (const R *) v.p0; o.p1 = (const R *) v.p1; o.p2 = (const R *) v.p2;
        o.n0 = (const R *) v.n0; o.n1 = (const R *) v.n1; o.n2 = (const R *) v.n2;
        o.b0 = (const R *) v.b0; o.b1 = (const R *) v.b1; o.eta = (const R *) v.eta;
        o.light = (const R *) v.light;
        o.bsdf = v.bsdf; o.active = v.active; o.active_em = v.active_em; o.ismesh = v.ismesh;
    }
    A.dlduv = (const R *) dlduv; A.stride = stride; A.dcols = dcols > 2 * K ? 2 * K : dcols;
    A.dldp = (const R *) dldp;
    A.clip = (clip > 0 && clip < 1e300) ? (R) clip : realmax_(R(0));
    A.op = (R *) op; A.ol = (R *) ol; A.od = (R *) od;
    memset(op, 0, sizeof(R) * 3 * (size_t) N * (size_t) A.P);
    memset(ol, 0, sizeof(R) * 3 * (size_t) N * (size_t) K);
    memset(od, 0, sizeof(R) * 3 * (size_t) N * (size_t) K);
    for (int64_t i = 0; i < N; ++i) {
        if (variant == EPSM_VARIANT_MANIFOLD) manifold_one(A, i); else caustic_one(A, i);
    }
    return 0;
}

extern "C" int epsm_host_cp_grad_f32(int variant, int64_t N, int K, const void *cam, const EpsmVertexRecord *verts,
                                     const void *dlduv, int64_t stride, int dcols, const void *dldp, double clip,
                                     void *op, void *ol, void *od, int) {
    return run<float>(variant, N, K, cam, verts, dlduv, stride, dcols, dldp, clip, op, ol, od);
}
extern "C" int epsm_host_cp_grad_f64(int variant, int64_t N, int K, const void *cam, const EpsmVertexRecord *verts,
                                     const void *dlduv, int64_t stride, int dcols, const void *dldp, double clip,
                                     void *op, void *ol, void *od, int) {
    return run<double>(variant, N, K, cam, verts, dlduv, stride, dcols, dldp, clip, op, ol, od);
}

// the term masks of a flag word (cp::manifold_plan / caustic_plan) and the tracer's retirement rule (cp::gradient_live)
extern "C" uint32_t epsm_host_plan(int variant, uint32_t w) { return variant == 0 ? cp::manifold_plan(w) : cp::caustic_plan(w); }
extern "C" int epsm_host_gradient_live(uint32_t w, int k, int caustic) { return cp::gradient_live(w, k, caustic != 0) ? 1 : 0; }
