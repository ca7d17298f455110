// epsm_scatter_core.h -- per-path adjoint of the scene-parameter gathers.
//
// What the reference obtains by replaying the whole path trace in Backward mode
// and letting Dr.Jit differentiate  Mesh::vertex_position / vertex_normal
// (include/mitsuba/render/mesh.h:94-106) and the BSDF sample
// (epsm.py:283-297 -> 559-562, 622-627, 644-645) is, per logged vertex, a fixed
// linear map of calc_grad's outputs into three flat buffers.  `vertex_items`
// evaluates that map for one (path, vertex): up to 3 position rows, 3 normal
// rows, 1 alpha slot and 3 emitter-triangle rows.  How the items are summed is
// the caller's business: float atomics after a wave-level merge on the GPU
// (epsm_scatter.hip), plain fp64 += in the oracle.
#pragma once

#include "epsm_path_core.h"

namespace epsm {

constexpr uint32_t kNoIndex = 0xFFFFFFFFu;
constexpr uint32_t kModeVertexNormals = 0x1u, kModeFlip = 0x2u, kModePos = 0x4u, kModeNrm = 0x8u;

// EpsmScatterRecord (include/epsm.h): a triangle id and up to three packed 16-byte records per vertex.
// Float fields travel as raw bits; `R` only matters for the host harness / oracle where the
// arrays still hold fp32.
template <typename R> struct ScatterPtrs {
    const uint32_t *tri;       // (N)   id of the hit triangle (row of the triangle table)
    const uint32_t *aux;       // (N,4) bsdf_id, dhf xyz   or null
    const uint32_t *emit;      // (N,4) etri, eb0, eb1, ew   or null
    const uint32_t *shadow;    // (N,4) stri, sb0, sb1, dis   or null; first vertex only
};
// The scene's triangle table (include/epsm.h): row t = [v0, v1, v2, mode]; ids >= T address nothing.
struct TriTable { const uint32_t *rows; int64_t T; };
EPSM_HD float bits_to_float(uint32_t u) { union { uint32_t u; float f; } c; c.u = u; return c.f; }
struct U4 { uint32_t x, y, z, w; };
EPSM_HD U4 load_u4(const uint32_t *base, int64_t i) {
#if defined(__HIP_DEVICE_COMPILE__)
    // one 16-byte GLOBAL load (gl(), epsm_path_core.h; a built-in vector type: copying a HIP uint4 out of an
    // address-space pointer goes through a generic reference and becomes a flat load again)
    typedef uint32_t U32x4 __attribute__((ext_vector_type(4)));
    const U32x4 v = *(const __attribute__((address_space(1))) U32x4 *) (base + 4 * i);
    U4 o; o.x = v.x; o.y = v.y; o.z = v.z; o.w = v.w; return o;
#else
    const uint32_t *p = base + 4 * i;
    U4 o; o.x = p[0]; o.y = p[1]; o.z = p[2]; o.w = p[3]; return o;
#endif
}

// one row of the triangle table; ids beyond the table give "no triangle"
EPSM_HD U4 table_row(const TriTable &tab, uint32_t id) {
    U4 r; r.x = r.y = r.z = 0xFFFFFFFFu; r.w = 0u;
    if ((int64_t) id < tab.T) r = load_u4(tab.rows, (int64_t) id);
    return r;
}

template <typename R> struct ScatterArgs {
    int64_t N;
    int K, P;                  // P = number of (N,3) arrays in out_param (5K or 5K-2)
    VertexPtrs<R> v[kMaxVertices];
    ScatterPtrs<R> s[kMaxVertices];
    TriTable tab;
    const R *out_param, *out_light, *out_diffuse;
    int64_t V, B;
};

template <typename R> EPSM_HD bool nz3(V3<R> g) { return g.x != R(0) || g.y != R(0) || g.z != R(0); }
template <typename R> EPSM_HD V3<R> load_out(const R *base, int64_t slot, int64_t N, int64_t i) {
    return load3(base + slot * N * 3, i);
}

// The gradients calc_grad produced for vertex k of one path.
template <typename R> struct VertexGrads {
    V3<R> gp[3];     // d/d p0,p1,p2      out_param[5it+0..2]
    V3<R> gn, gm;    // d/d n, d/d m      out_param[5it+3], [5it+4]
    V3<R> glight;    // light_grad[it]
    V3<R> gdiff;     // diffuse_grad[it]
    bool has_nm;     // epsm.py:559,644: `iteration*5+4 < len(path_grad)`
};
template <typename R> EPSM_HD VertexGrads<R> load_vertex_grads(const ScatterArgs<R> &A, int64_t i, int it) {
    VertexGrads<R> g;
    g.has_nm = 5 * it + 4 < A.P;
    for (int j = 0; j < 3; ++j) g.gp[j] = g.has_nm ? load_out(A.out_param, 5 * it + j, A.N, i) : zero3<R>();
    g.gn = g.has_nm ? load_out(A.out_param, 5 * it + 3, A.N, i) : zero3<R>();
    g.gm = g.has_nm ? load_out(A.out_param, 5 * it + 4, A.N, i) : zero3<R>();
    g.glight = load_out(A.out_light, it, A.N, i);
    g.gdiff = load_out(A.out_diffuse, it, A.N, i);
    return g;
}

template <typename R> struct VertexItems {
    uint32_t vi[3];  bool pos_ok, nrm_ok;     // hit triangle (valid indices && attached)
    V3<R> pos[3], nrm[3];
    uint32_t bid;    bool alpha_ok;  R alpha;
    uint32_t ei[3];  bool em_ok;     V3<R> em[3];
    uint32_t si[3];  bool sh_ok;     V3<R> sh[3];     // occluder of the first vertex's emitter sample (epsm.py:609-620)
};

// Occluder term: si_direct.p * diffuse_grad[0] * dis with detached barycentrics (epsm.py:609-620).
template <typename R> struct ShadowItems { uint32_t si[3]; bool ok; V3<R> val[3]; };
template <typename R>
EPSM_HD ShadowItems<R> shadow_items(const uint32_t *shadow, const TriTable &tab, int64_t i, V3<R> gdiff, int64_t V) {
    ShadowItems<R> o;
    o.ok = false; o.si[0] = o.si[1] = o.si[2] = kNoIndex; o.val[0] = o.val[1] = o.val[2] = zero3<R>();
    if (!shadow || !nz3(gdiff)) return o;
    const U4 a = load_u4(shadow, i);
    const U4 row = table_row(tab, a.x);
    o.si[0] = row.x; o.si[1] = row.y; o.si[2] = row.z;
    const R c0 = R(bits_to_float(a.y)), c1 = R(bits_to_float(a.z)), dis = R(bits_to_float(a.w));
    const uint32_t mode = row.w;
    if (o.si[0] < (uint64_t) V && o.si[1] < (uint64_t) V && o.si[2] < (uint64_t) V && (mode & kModePos) && dis != R(0)) {
        o.ok = true;
        const V3<R> g = gdiff * dis;
        o.val[0] = g * c0; o.val[1] = g * c1; o.val[2] = g * (R(1) - c0 - c1);
    }
    return o;
}

template <typename R>
EPSM_HD VertexItems<R> vertex_items(const VertexPtrs<R> &v, const ScatterPtrs<R> &s, const TriTable &tab, int64_t i,
                                    const VertexGrads<R> &g, int64_t V, int64_t B) {
    VertexItems<R> o;
    const U4 t4 = table_row(tab, gl(s.tri)[i]);
    const uint32_t mode = t4.w;
    o.vi[0] = t4.x; o.vi[1] = t4.y; o.vi[2] = t4.z;
    const bool idx_ok = o.vi[0] < (uint64_t) V && o.vi[1] < (uint64_t) V && o.vi[2] < (uint64_t) V;
    o.pos_ok = idx_ok && (mode & kModePos);
    o.nrm_ok = idx_ok && (mode & kModeNrm) && (mode & kModeVertexNormals);
    const R b0 = gl(v.b0)[i], b1 = gl(v.b1)[i], b2 = R(1) - b0 - b1;
    // (1) si.p_j * path_grad[5it+j] (epsm.py:559-560)  +  (2) si_follow.p * diffuse_grad[it] with
    //     detached barycentrics (epsm.py:561-562)
    o.pos[0] = g.gp[0] + g.gdiff * b0;
    o.pos[1] = g.gp[1] + g.gdiff * b1;
    o.pos[2] = g.gp[2] + g.gdiff * b2;
    o.nrm[0] = o.nrm[1] = o.nrm[2] = zero3<R>();
    // (3) si_follow.sh_frame.n * path_grad[5it+3]  (epsm.py:644-645)
    if (g.has_nm && nz3(g.gn)) {
        const R sgn = (mode & kModeFlip) ? R(-1) : R(1);
        if (mode & kModeVertexNormals) {
            if (o.nrm_ok) {
                // logged normals are post-flip (mesh.cpp:820-827): n' = sgn * buffer value;
                // sh = normalize(sum b_j n'_j); d(sh.g)/d n_j = sgn b_j (g - sh (sh.g)) / |n'|
                const V3<R> n = load3(v.n0, i) * b0 + load3(v.n1, i) * b1 + load3(v.n2, i) * b2;
                const R il = rsqrt_(dot(n, n));
                const V3<R> sh = n * il;
                const V3<R> pg = (g.gn - sh * dot(sh, g.gn)) * (il * sgn);
                o.nrm[0] = pg * b0; o.nrm[1] = pg * b1; o.nrm[2] = pg * b2;
            }
        } else if (o.pos_ok) {
            // flat mesh: sh = sgn * normalize(cross(p1-p0, p2-p0))  (mesh.cpp:729,811)
            const V3<R> p0 = load3(v.p0, i), p1 = load3(v.p1, i), p2 = load3(v.p2, i);
            const V3<R> d0 = p1 - p0, d1 = p2 - p0;
            const V3<R> c = cross(d0, d1);
            const R il = rsqrt_(dot(c, c));
            const V3<R> ch = c * il;
            const V3<R> cb = (g.gn - ch * dot(ch, g.gn)) * (il * sgn);
            const V3<R> d0b = cross(d1, cb), d1b = cross(cb, d0);
            o.pos[1] = o.pos[1] + d0b; o.pos[2] = o.pos[2] + d1b; o.pos[0] = o.pos[0] - (d0b + d1b);
        }
    }
    // bsdf_sample.hf * path_grad[5it+4]  (epsm.py:645)
    o.alpha_ok = false; o.alpha = R(0); o.bid = kNoIndex;
    if (g.has_nm && s.aux) {
        const U4 a4 = load_u4(s.aux, i);
        o.bid = a4.x;
        if (o.bid < (uint64_t) B && nz3(g.gm)) {
            o.alpha_ok = true;
            o.alpha = dot(g.gm, mk3<R>(R(bits_to_float(a4.y)), R(bits_to_float(a4.z)), R(bits_to_float(a4.w))));
        }
    }
    // (4) si_direct.p * light_grad[it] * sum(Lr_dir)  (epsm.py:622-627)
    o.em_ok = false; o.em[0] = o.em[1] = o.em[2] = zero3<R>(); o.ei[0] = o.ei[1] = o.ei[2] = kNoIndex;
    if (s.emit && nz3(g.glight)) {
        const U4 e4 = load_u4(s.emit, i);
        const U4 er = table_row(tab, e4.x);
        o.ei[0] = er.x; o.ei[1] = er.y; o.ei[2] = er.z;
        // (si_direct.p is AD-attached only when the emitter mesh's positions are)
        if (o.ei[0] < (uint64_t) V && o.ei[1] < (uint64_t) V && o.ei[2] < (uint64_t) V && (er.w & kModePos)) {
            const V3<R> gl = g.glight * R(bits_to_float(e4.w));
            const R c0 = R(bits_to_float(e4.y)), c1 = R(bits_to_float(e4.z));
            o.em_ok = true;
            o.em[0] = gl * c0; o.em[1] = gl * c1; o.em[2] = gl * (R(1) - c0 - c1);
        }
    }
    // (5) occluder of the emitter sample of the first vertex (epsm.py:609-620)
    const ShadowItems<R> sh = shadow_items<R>(s.shadow, tab, i, g.gdiff, V);
    o.sh_ok = sh.ok;
    for (int j = 0; j < 3; ++j) { o.si[j] = sh.si[j]; o.sh[j] = sh.val[j]; }
    return o;
}

}  // namespace epsm
