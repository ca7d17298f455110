// epsm_cp_core.h -- the manifold-gradient arithmetic in CONSTRAINT-PARALLEL form.
//
// epsm_path_core.h gives one lane a whole light path: per-vertex state of up to five vertices lives in registers between
// a forward and a backward pass (130 of the 256 VGPRs of the fused kernel), every half-vector constraint is evaluated
// twice and its frame three times.  Here the unit of work is ONE CONSTRAINT VERTEX of one path:
//
//   blocks    lane (path, k) loads vertex k and the positions / edges of its two neighbours, evaluates the projected
//             half-vector constraint(s) of vertex k with unit seeds and keeps the 2x2 blocks of `cur` (epsm.py:771,
//             826-833, 889-900) they contract to -- 20 numbers; the 2 x 12 Jacobians themselves are dropped;
//   recursion the block recursions of epsm_path_core.h (manifold: block LU forward, adjoint seeds backward;
//             manifold_caustic: one forward recursion) run ACROSS the lanes of a path on the blocks alone: step s is
//             taken by the lanes with k == s, which receive ten-odd numbers from lane k-1 (two from k+1);
//   contract  the lane's constraint(s) swept once more, SEEDED with the 2-vectors the recursion produced: every output
//             of vertex k at once, plus d/dx_k through constraint k+1, handed down by the neighbouring lane.
// (Round 3 kept the unit-seed Jacobians of both versions across the recursions -- 36 registers -- so that nothing was
// evaluated twice; the second sweep costs ~80 multiply-adds per version and is what lets three waves share a SIMD.)
//
// Per-lane state does not grow with the chain length, chains of different lengths keep all lanes busy, and the
// vertex loops are real loops (the per-vertex arrays that forced compile-time unrolling are gone).
//
// What is computed is what epsm_path_core.h computes -- ManifoldIntegrator.calc_grad (epsm.py:745-946) and
// ManifoldCausticIntegrator.calc_grad (epsm.py:952-1200) in adjoint form, same masks, same quirks, same NaN and clamp
// rules; the derivation is in that file's header and in DESIGN.md section 2.  The frame derivative is written in closed
// form here (epsm.py:746-756 differentiated by hand) instead of as the reverse sweep of the cross products.
//
// Plain C++ templated on the scalar type: hipcc compiles it into the gfx950 backward kernel (epsm_backward_cp.hip, lanes
// = wave lanes, exchange = shuffles) and tests/host_harness compiles the SAME functions for the CPU (lanes = array
// entries), so the algebra is checked against the reference's goldens and the oracle without a GPU.
#pragma once

#include "epsm_path_core.h"

namespace epsm {
namespace cp {

// ----------------------------------------------------------------------------
// plan of a path: which terms exist (from the flag word alone)
// ----------------------------------------------------------------------------
// Flag word: 5 bits per logged vertex (include/epsm.h EPSM_FLAG_*), vertex k at bits 5(k-1)..; vertices beyond K are 0.
// Plan word:
//   bits 0..4   manifold: wN[k] (light-sampling term at depth k)      caustic: wP[k] (epsm.py:1172-1174)
//   bits 5..9   manifold: wC[k] (continuing term at depth k)          caustic: wD[k] (epsm.py:1180-1182)
//   bits 10..12 nv: last vertex whose geometry is needed
//   bits 13..15 manifold: bit 13 = the last vertex has a light-sampling term      caustic: id* (0 = none)
//   bits 16..18 m: number of constraint vertices (lanes with a constraint); a path always gets max(m, 1) lanes,
//               the first of which also carries the first-vertex tangent and diffuse_grad[0]
//   bit  19     first logged vertex is diffuse
//   bit  20     path index inside the wavefront (set by the caller)
//   bit  21     first logged vertex is active (set by the caller: the first-vertex tangent needs it)
constexpr uint32_t kPlanInRange = 1u << 20, kPlanActive1 = 1u << 21;
EPSM_HD int plan_nv(uint32_t p) { return (int) ((p >> 10) & 7u); }
EPSM_HD int plan_m(uint32_t p) { return (int) ((p >> 16) & 7u); }
EPSM_HD int plan_idstar(uint32_t p) { return (int) ((p >> 13) & 7u); }
EPSM_HD bool plan_a(uint32_t p, int k) { return ((p >> (k - 1)) & 1u) != 0; }        // wN / wP
EPSM_HD bool plan_b(uint32_t p, int k) { return ((p >> (4 + k)) & 1u) != 0; }        // wC / wD
EPSM_HD bool plan_diffuse1(uint32_t p) { return ((p >> 19) & 1u) != 0; }

EPSM_HD bool fbit(uint32_t w, int k, uint32_t bit) { return k >= 1 && k <= 5 && ((w >> (5 * (k - 1))) & bit) != 0; }

// term masks of epsm.py:793-802, 852-855, 916-920 (manifold_path of epsm_path_core.h)
EPSM_HD uint32_t manifold_plan(uint32_t w) {
    bool valid = true;
    int hasdiffuse = 0, nv = 0;
    bool nvN = false;
    uint32_t a = 0, b = 0;
#pragma unroll
    for (int id = 1; id <= 5; ++id) {
        valid = valid && fbit(w, id, 16u);
        hasdiffuse += fbit(w, id, 1u) ? 1 : 0;
        valid = valid && (hasdiffuse < 2);
        const bool spec = valid && (hasdiffuse == 0);
        const bool wN = spec && fbit(w, id, 4u) && fbit(w, id, 8u);
        const bool wC = spec && fbit(w, id + 1, 4u) && fbit(w, id + 1, 1u);
        if (wN) { nv = id; nvN = true; a |= 1u << (id - 1); }
        if (wC) { nv = id + 1; nvN = false; b |= 1u << (id - 1); }
    }
    const int m = nvN ? nv : (nv > 0 ? nv - 1 : 0);
    return a | (b << 5) | ((uint32_t) nv << 10) | (nvN ? 1u << 13 : 0u) | ((uint32_t) m << 16) | (fbit(w, 1, 1u) ? 1u << 19 : 0u);
}
// epsm.py:1172-1183 (caustic_path of epsm_path_core.h)
EPSM_HD uint32_t caustic_plan(uint32_t w) {
    bool valid = true;
    int hasdiffuse = 0, nv = 0, idstar = 0;
    uint32_t a = 0, b = 0;
    const bool d1 = fbit(w, 1, 1u);
#pragma unroll
    for (int id = 1; id <= 5; ++id) {
        valid = valid && fbit(w, id, 16u);
        hasdiffuse += fbit(w, id, 1u) ? 1 : 0;
        valid = valid && (hasdiffuse < 2);
        const bool base = d1 && valid && fbit(w, id + 1, 4u);
        const bool wP = base && fbit(w, id + 1, 1u);
        const bool wD = base && (fbit(w, id + 1, 1u) || fbit(w, id + 1, 2u));
        if (wD) { nv = id + 1; b |= 1u << (id - 1); }
        if (wP) { idstar = id; a |= 1u << (id - 1); }
    }
    const int m = nv > 0 ? nv - 1 : 0;
    return a | (b << 5) | ((uint32_t) nv << 10) | ((uint32_t) idstar << 13) | ((uint32_t) m << 16) | (d1 ? 1u << 19 : 0u);
}

// Can a vertex AFTER vertex k still produce, or be read by, a term?  `w` = the flag word of vertices 1..k.  The masks above:
// every term at depth id needs `valid` (vertices 1..id mesh hits, fewer than two of them Diffuse) and, for manifold,
// hasdiffuse == 0 over 1..id; a term at id < k reads vertices <= id + 1 <= k.  So once the condition fails at k it fails
// for every id >= k and nothing behind vertex k is ever looked at (the tracer's EPSM_TRACE_GRADIENT_ONLY retires the path).
EPSM_HD bool gradient_live(uint32_t w, int k, bool caustic) {
    bool all_mesh = true;
    int ndiffuse = 0;
    for (int j = 1; j <= 5; ++j)
        if (j <= k) { all_mesh = all_mesh && fbit(w, j, 16u); ndiffuse += fbit(w, j, 1u) ? 1 : 0; }
    return caustic ? (fbit(w, 1, 1u) && all_mesh && ndiffuse < 2) : (all_mesh && ndiffuse == 0);
}

// ----------------------------------------------------------------------------
// geometry a lane holds
// ----------------------------------------------------------------------------
template <typename R> struct Own {      // vertex k itself
    V3<R> x, e1, e2;                    // x = sum b_j p_j, e_j = dx/db_j   (epsm.py:758-759)
    V3<R> n, dn1, dn2;                  // un-normalised interpolated normal and dn/db_j   (epsm.py:761-762)
    R b0, b1, eta;
    V3<R> light;                        // emitter sample point
};
template <typename R> struct Nbr { V3<R> x, e1, e2; };      // vertex k-1 / k+1 (the camera: e1 = e2 = 0)

// local frame of epsm.py:746-756 in closed form: t = (0, ty, tz) = (0, -nn_z, nn_y) / sg,  bt = nn x t
template <typename R> struct Frm { V3<R> nn, bt; R inv_n, ty, tz, isg; };
template <typename R> EPSM_HD Frm<R> make_frm(V3<R> n) {
    Frm<R> f;
    f.inv_n = rsqrt_(dot(n, n));
    f.nn = n * f.inv_n;
    const R s2 = f.nn.y * f.nn.y + f.nn.z * f.nn.z;      // n || x: 0 -> isg non-finite -> the term is dropped (epsm.py:746-748, 856)
    f.isg = rsqrt_(s2);
    f.ty = -f.nn.z * f.isg;
    f.tz = f.nn.y * f.isg;
    f.bt = mk3<R>(s2 * f.isg, -f.nn.x * f.tz, f.nn.x * f.ty);
    return f;
}
// d(t . w)/dn and d(bt . w)/dn for a vector w that does not depend on n (rows of the frame differentiated by hand,
// then through nn = n / |n|)
template <typename R> EPSM_HD void frame_grad(const Frm<R> &f, V3<R> w, R cx, V3<R> &g0, V3<R> &g1) {
    const R x = f.nn.x, q = f.nn.y * w.y + f.nn.z * w.z, xi = x * f.isg, qi = q * f.isg;
    const V3<R> G0 = mk3<R>(R(0), (w.z - cx * f.tz) * f.isg, (cx * f.ty - w.y) * f.isg);
    const V3<R> G1 = mk3<R>(-qi, w.x * f.tz + xi * (qi * f.tz - w.y), -w.x * f.ty - xi * (w.z + qi * f.ty));
    g0 = (G0 - f.nn * dot(f.nn, G0)) * f.inv_n;
    g1 = (G1 - f.nn * dot(f.nn, G1)) * f.inv_n;
}

// Unit-seed reverse sweeps of one constraint: row r = gradient of component r.  d/dx_k = -(gxp + gxn).
template <typename R> struct Jac { V3<R> gxp[2], gxn[2], gn[2]; };
template <typename R> EPSM_HD Jac<R> zero_jac() {
    Jac<R> j;
    j.gxp[0] = j.gxp[1] = j.gxn[0] = j.gxn[1] = j.gn[0] = j.gn[1] = zero3<R>();
    return j;
}
template <typename R> struct Dir { V3<R> w; R inv; };      // unit direction and 1/length
template <typename R> EPSM_HD Dir<R> make_dir(V3<R> from, V3<R> to) {
    Dir<R> d;
    const V3<R> a = to - from;
    d.inv = rsqrt_(dot(a, a));
    d.w = a * d.inv;
    return d;
}
// C = [ normalize(R wi + eta R wo) ]_xy   (epsm.py:809-821)
template <typename R> EPSM_HD Jac<R> halfvec_jac(const Frm<R> &f, const Dir<R> &wi, const Dir<R> &wo, R eta) {
    Jac<R> j;
    const V3<R> u = madd(wi.w, wo.w, eta);
    const R inv_u = rsqrt_(dot(u, u));
    const V3<R> uh = u * inv_u;
    const R cx = f.ty * uh.y + f.tz * uh.z, cy = dot(f.bt, uh);
    // d C_r / d u
    const V3<R> P0 = (mk3<R>(R(0), f.ty, f.tz) - uh * cx) * inv_u, P1 = (f.bt - uh * cy) * inv_u;
    j.gxp[0] = (P0 - wi.w * dot(wi.w, P0)) * wi.inv;
    j.gxp[1] = (P1 - wi.w * dot(wi.w, P1)) * wi.inv;
    const R eb = eta * wo.inv;
    j.gxn[0] = (P0 - wo.w * dot(wo.w, P0)) * eb;
    j.gxn[1] = (P1 - wo.w * dot(wo.w, P1)) * eb;
    frame_grad(f, uh, cx, j.gn[0], j.gn[1]);
    return j;
}
// pseudo-constraint of manifold_caustic: wo2 = [R normalize(x_next - x)]_xy   (epsm.py:1028, 1116); gxp = 0
template <typename R> EPSM_HD Jac<R> wo2_jac(const Frm<R> &f, const Dir<R> &wo) {
    Jac<R> j;
    const R cx = f.ty * wo.w.y + f.tz * wo.w.z, cy = dot(f.bt, wo.w);
    j.gxp[0] = j.gxp[1] = zero3<R>();
    j.gxn[0] = (mk3<R>(R(0), f.ty, f.tz) - wo.w * cx) * wo.inv;
    j.gxn[1] = (f.bt - wo.w * cy) * wo.inv;
    frame_grad(f, wo.w, cx, j.gn[0], j.gn[1]);
    return j;
}
template <typename R> EPSM_HD V3<R> gxc(const Jac<R> &j, int r) { return -(j.gxp[r] + j.gxn[r]); }

// seeded combination of the two rows
template <typename R> EPSM_HD V3<R> lin(const V3<R> v[2], V2<R> s) { return v[0] * s.x + v[1] * s.y; }

// 2x2 blocks of `cur`: rows = components of the constraint, columns = (b0, b1) of a vertex
template <typename R> EPSM_HD M2<R> blk(const V3<R> g[2], V3<R> e1, V3<R> e2) { return block2(g[0], g[1], e1, e2); }
template <typename R> EPSM_HD M2<R> blk_own(const Jac<R> &j, const Own<R> &o) {
    const V3<R> c0 = gxc(j, 0), c1 = gxc(j, 1);
    return madd2(block2(c0, c1, o.e1, o.e2), block2(j.gn[0], j.gn[1], o.dn1, o.dn2));
}

// ----------------------------------------------------------------------------
// seeded reverse sweeps: s . J for a 2-vector s, without the unit-seed rows (s . gxp, s . gxn, s . gn of the Jac above)
// ----------------------------------------------------------------------------
template <typename R> struct Swp { V3<R> p, n, g; };      // d/dx_{k-1}, d/dx_{k+1}, d/dn_k;  d/dx_k = -(p + n)
template <typename R> EPSM_HD Swp<R> zero_swp() { Swp<R> w; w.p = w.n = w.g = zero3<R>(); return w; }
template <typename R> EPSM_HD V3<R> frame_grad_seeded(const Frm<R> &f, V3<R> w, R cx, V2<R> s) {
    const R x = f.nn.x, q = f.nn.y * w.y + f.nn.z * w.z, xi = x * f.isg, qi = q * f.isg;
    const V3<R> G0 = mk3<R>(R(0), (w.z - cx * f.tz) * f.isg, (cx * f.ty - w.y) * f.isg);
    const V3<R> G1 = mk3<R>(-qi, w.x * f.tz + xi * (qi * f.tz - w.y), -w.x * f.ty - xi * (w.z + qi * f.ty));
    const V3<R> G = G0 * s.x + G1 * s.y;
    return (G - f.nn * dot(f.nn, G)) * f.inv_n;
}
template <typename R> EPSM_HD Swp<R> halfvec_seeded(const Frm<R> &f, const Dir<R> &wi, const Dir<R> &wo, R eta, V2<R> s) {
    Swp<R> o;
    const V3<R> u = madd(wi.w, wo.w, eta);
    const R inv_u = rsqrt_(dot(u, u));
    const V3<R> uh = u * inv_u;
    const R cx = f.ty * uh.y + f.tz * uh.z, cy = dot(f.bt, uh);
    const V3<R> P = (mk3<R>(R(0), f.ty, f.tz) * s.x + f.bt * s.y - uh * (cx * s.x + cy * s.y)) * inv_u;      // s . dC/du
    o.p = (P - wi.w * dot(wi.w, P)) * wi.inv;
    o.n = (P - wo.w * dot(wo.w, P)) * (eta * wo.inv);
    o.g = frame_grad_seeded(f, uh, cx, s);
    return o;
}
template <typename R> EPSM_HD Swp<R> wo2_seeded(const Frm<R> &f, const Dir<R> &wo, V2<R> s) {      // p == 0
    Swp<R> o;
    const R cx = f.ty * wo.w.y + f.tz * wo.w.z, cy = dot(f.bt, wo.w);
    o.p = zero3<R>();
    o.n = (mk3<R>(R(0), f.ty, f.tz) * s.x + f.bt * s.y - wo.w * (cx * s.x + cy * s.y)) * wo.inv;
    o.g = frame_grad_seeded(f, wo.w, cx, s);
    return o;
}

// What a lane keeps of its vertex between the two passes: positions only (directions and the frame are rebuilt: five
// v_rsq_f32 and ~70 multiply-adds against 23 registers held across the recursions).
template <typename R> struct Pts { V3<R> x, n, light, xp, xn; R eta; };      // x_k, n_k, emitter sample, x_{k-1}, x_{k+1}

// ============================================================================
// "manifold"
// ============================================================================
// Pass 1: the 2x2 blocks of `cur` the lane's constraint(s) contribute (epsm.py:826-833, 889-900) -- the unit-seed sweeps they
// are contracted from are NOT kept (round 3 kept both versions' 2 x 12 Jacobians, 36 registers, across the recursions).
template <typename R> struct MBlocks { M2<R> AkkN, AkmN, AkkC, AkmC, Aup; };   // A_{k,k}, A_{k,k-1} of both versions; A^C_{k,k+1}
// lane (path, k): wN = plan_a(k), has_next = k + 1 <= nv
template <typename R> EPSM_HD MBlocks<R> manifold_blocks(const Own<R> &o, const Nbr<R> &prev, const Nbr<R> &next, bool wN, bool has_next) {
    MBlocks<R> e;
    const M2<R> z{R(0), R(0), R(0), R(0)};
    e.AkkN = e.AkmN = e.AkkC = e.AkmC = e.Aup = z;
    const Frm<R> f = make_frm(o.n);
    const Dir<R> wi = make_dir(o.x, prev.x);
    if (wN) {
        const Jac<R> j = halfvec_jac(f, wi, make_dir(o.x, o.light), o.eta);
        e.AkkN = blk_own(j, o);
        e.AkmN = blk(j.gxp, prev.e1, prev.e2);
    }
    if (has_next) {
        const Jac<R> j = halfvec_jac(f, wi, make_dir(o.x, next.x), o.eta);
        e.AkkC = blk_own(j, o);
        e.AkmC = blk(j.gxp, prev.e1, prev.e2);
        e.Aup = blk(j.gxn, next.e1, next.e2);
    }
    return e;
}

// what lane k hands to lane k+1 in the forward recursion / keeps for the backward one
template <typename R> struct MFwd { V2<R> z, zN; M2<R> Sinv; };
template <typename R> EPSM_HD MFwd<R> mfwd_zero() {
    MFwd<R> f;
    f.z = f.zN = mk2<R>(R(0), R(0));
    f.Sinv = M2<R>{R(0), R(0), R(0), R(0)};
    return f;
}
// forward step of lane k (block LU of the continuing rows; pass 1 of manifold_path).  `pf`, `pAup`: z, Sinv and
// A^C_{k-1,k} of lane k-1 (ignored for k == 1).  Computed in RecType (float64, see epsm_path_core.h), kept in R.
template <typename R>
EPSM_HD MFwd<R> manifold_fwd(const MBlocks<R> &e, V2<R> dk, bool first, const MFwd<R> &pf, const M2<R> &pAup, bool wN, bool has_next) {
    typedef typename RecType<R>::type Q;
    MFwd<R> o = mfwd_zero<R>();
    V2<Q> rhs = cvv<Q>(dk);
    M2<Q> T{Q(0), Q(0), Q(0), Q(0)};
    if (!first) {
        const M2<Q> Au = cvm<Q>(pAup);
        rhs = rhs - vmul(cvv<Q>(pf.z), Au);
        T = mmul(cvm<Q>(pf.Sinv), Au);
    }
    if (wN) {
        M2<Q> S = cvm<Q>(e.AkkN);
        if (!first) S = msub(S, mmul(cvm<Q>(e.AkmN), T));
        o.zN = cvv<R>(vmul(rhs, minv(S)));
    }
    if (has_next) {
        M2<Q> S = cvm<Q>(e.AkkC);
        if (!first) S = msub(S, mmul(cvm<Q>(e.AkmC), T));
        const M2<Q> Si = minv(S);
        o.Sinv = cvm<R>(Si);
        o.z = cvv<R>(vmul(rhs, Si));
    }
    return o;
}
// What lane k hands to lane k-1 in the backward recursion: q = (GP . e1, GP . e2) of vertex k-1, GP = d/dx_{k-1} through
// constraint k with the lane's final seeds -- which is sN A^N_{k,k-1} + sC A^C_{k,k-1}, so the recursion runs on the blocks
// alone -- and the number W of live terms of depth >= k.
template <typename R> struct MBwd { V2<R> q; int W; };
// the adjoint seeds of lane k: y_k summed over the depths and sub-paths that reach it
template <typename R> struct MSeeds { V2<R> sN, sC; bool useN, useC, fC; };
// backward step of lane k (pass 2 of manifold_path): `nb` comes from lane k+1 (zero for the last lane)
template <typename R>
EPSM_HD MSeeds<R> manifold_bwd(const MBlocks<R> &e, const MFwd<R> &f, const MBwd<R> &nb, bool wN, bool wC, bool has_next, MBwd<R> &mine) {
    typedef typename RecType<R>::type Q;
    MSeeds<R> s;
    // carry = W z_k - (GP . [e1 e2]_k) Sinv_k : the sum over deeper terms of their y_k
    V2<Q> carry = mk2<Q>(Q(0), Q(0));
    if (nb.W > 0) carry = cvv<Q>(f.z) * Q(nb.W) - vmul(cvv<Q>(nb.q), cvm<Q>(f.Sinv));
    const bool fN = wN && finite2(f.zN);
    s.fC = wC && finite2(f.z);
    s.sN = fN ? f.zN : mk2<R>(R(0), R(0));
    V2<Q> sCq = carry;
    if (s.fC) sCq = sCq + cvv<Q>(f.z);
    s.sC = cvv<R>(sCq);
    s.useN = s.sN.x != R(0) || s.sN.y != R(0);
    s.useC = has_next && (s.sC.x != R(0) || s.sC.y != R(0));
    V2<Q> q = mk2<Q>(Q(0), Q(0));
    if (s.useN) q = q + vmul(cvv<Q>(s.sN), cvm<Q>(e.AkmN));
    if (s.useC) q = q + vmul(sCq, cvm<Q>(e.AkmC));
    mine.q = cvv<R>(q);
    mine.W = nb.W + (fN ? 1 : 0) + (s.fC ? 1 : 0);
    return s;
}
// results of lane k.  Gx lacks d/dx_k through constraint k+1: the caller subtracts GP of lane k+1.
template <typename R> struct MOut { V3<R> Gx, gn, gm, glight, gdiff, GP; };
// Pass 2: both versions swept once more, seeded with the lane's final seeds.
template <typename R> EPSM_HD MOut<R> manifold_contract(const Pts<R> &p, const MSeeds<R> &s, bool has_next) {
    MOut<R> out;
    const V3<R> z3 = zero3<R>();
    Swp<R> a = zero_swp<R>(), c = a;
    if (s.useN || s.useC) {
        const Frm<R> f = make_frm(p.n);
        const Dir<R> wi = make_dir(p.x, p.xp);
        if (s.useN) a = halfvec_seeded(f, wi, make_dir(p.x, p.light), p.eta, s.sN);
        if (s.useC) c = halfvec_seeded(f, wi, make_dir(p.x, p.xn), p.eta, s.sC);
    }
    out.Gx = (a.p + a.n) + (c.p + c.n);                       // -(a.gxc + c.gxc) with gxc = -(gxp + gxn)
    out.gn = -(a.g + c.g);
    out.gm = has_next ? mk3<R>(s.sC.x, s.sC.y, R(0)) : z3;    // dC/dm = -I on continuing rows (epsm.py:883)
    out.glight = -a.n;
    out.gdiff = s.fC ? -c.n : z3;                             // carry == 0 whenever fC (a diffuse x_{k+1} ends the chain)
    out.GP = a.p + c.p;
    return out;
}

// ============================================================================
// "manifold_caustic"
// ============================================================================
template <typename R> struct CBlocks { M2<R> Akk, Akm, Aup, Wk; };      // half-vector constraint of vertex k (k >= 2); pseudo-constraint wo2 of depth k
template <typename R> EPSM_HD CBlocks<R> caustic_blocks(const Own<R> &o, const Nbr<R> &prev, const Nbr<R> &next, bool first) {
    CBlocks<R> e;
    const M2<R> z{R(0), R(0), R(0), R(0)};
    const Frm<R> f = make_frm(o.n);
    const Dir<R> wo = make_dir(o.x, next.x);
    e.Wk = blk_own(wo2_jac(f, wo), o);
    e.Akk = e.Akm = e.Aup = z;
    if (!first) {
        const Jac<R> j = halfvec_jac(f, make_dir(o.x, prev.x), wo, o.eta);
        e.Akm = blk(j.gxp, prev.e1, prev.e2);
        e.Akk = blk_own(j, o);
        e.Aup = blk(j.gxn, next.e1, next.e2);
    }
    return e;
}
// forward recursion: v_k = r_{k-1} A_{k,k-1}^-1, r_k = d_k - v_k A_kk - v_{k-1} A_{k-1,k}, u_k = r_k W_k^-1
template <typename R> struct CFwd { V2<R> v, r, u; bool fin; };
template <typename R> EPSM_HD CFwd<R> cfwd_zero() {
    CFwd<R> f;
    f.v = f.r = f.u = mk2<R>(R(0), R(0));
    f.fin = true;
    return f;
}
template <typename R>
EPSM_HD CFwd<R> caustic_fwd(const CBlocks<R> &e, V2<R> dk, bool first, const CFwd<R> &pf, const M2<R> &pAup) {
    CFwd<R> o;
    o.v = mk2<R>(R(0), R(0));
    V2<R> rk = dk;
    if (!first) {
        o.v = vmul(pf.r, minv(e.Akm));
        rk = dk - vmul(o.v, e.Akk) - vmul(pf.v, pAup);
    }
    o.r = rk;
    o.u = vmul(rk, minv(e.Wk));
    o.fin = finite2(o.u) && finite2(o.v);
    return o;
}
template <typename R> struct COut { V3<R> Gx, gn, gm, gdiff, gxp_prev; };
// lane k once the recursion has reached it (pass 2: seeded sweeps); `inP` = k <= id*, `star` = k == id*, wD = plan_b(k).
// gxp_prev: -(d/dx_{k-1}) through constraint k, to be ADDED to Gx of lane k-1.
template <typename R> EPSM_HD COut<R> caustic_finish(const Pts<R> &p, const CFwd<R> &f, bool first, bool inP, bool star, bool wD) {
    COut<R> o;
    const V3<R> z3 = zero3<R>();
    o.Gx = o.gn = o.gm = o.gdiff = o.gxp_prev = z3;
    const Frm<R> fr = make_frm(p.n);
    const Dir<R> wo = make_dir(p.x, p.xn);
    Swp<R> cs = zero_swp<R>();
    if (!first) cs = halfvec_seeded(fr, make_dir(p.x, p.xp), wo, p.eta, f.v);
    if (inP) {
        o.Gx = cs.p + cs.n;                       // -cs.gxc
        o.gn = -cs.g;
        if (!first) o.gm = mk3<R>(f.v.x, f.v.y, R(0));
        o.gxp_prev = -cs.p;
        if (star) {
            const Swp<R> ws = wo2_seeded(fr, wo, f.u);
            o.Gx = o.Gx + ws.n;                   // -ws.gxc, gxc = -gxn
            o.gn = o.gn - ws.g;
        }
    }
    if (wD && f.fin) {
        // epsm.py:1139-1157: row block id (gxn, plus the stale wo2[0] gradient on its second row) and the pseudo rows
        const V2<R> sw = first ? f.u : mk2<R>(f.u.x + f.v.y, f.u.y);
        const Swp<R> wx = wo2_seeded(fr, wo, sw);
        o.gdiff = first ? -wx.n : -(wx.n + cs.n);
    }
    return o;
}

}  // namespace cp
}  // namespace epsm
