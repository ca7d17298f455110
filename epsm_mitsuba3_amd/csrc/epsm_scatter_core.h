// epsm_scatter_core.h -- per-path adjoint of the scene-parameter gathers.
//
// What the reference obtains by replaying the whole path trace in Backward mode
// and letting Dr.Jit differentiate  Mesh::vertex_position / vertex_normal
// (include/mitsuba/render/mesh.h:94-106) and the BSDF sample
// (epsm.py:283-297 -> 559-562, 622-627, 644-645) is, per logged vertex, a fixed
// linear map of calc_grad's outputs into three flat buffers.  `Sink` adds one
// 3-vector / scalar into a buffer: float atomics on the GPU
// (epsm_scatter.hip), plain fp64 += in the host harness and the oracle.
#pragma once

#include "epsm_path_core.h"

namespace epsm {

constexpr uint32_t kNoIndex = 0xFFFFFFFFu;
constexpr uint32_t kModeVertexNormals = 0x1u, kModeFlip = 0x2u, kModePos = 0x4u, kModeNrm = 0x8u;

template <typename R> struct ScatterPtrs {
    const uint32_t *vidx;      // (N,3)
    const uint8_t *mode;       // (N)
    const uint32_t *bsdf_id;   // (N) or null
    const R *dhf_dalpha;       // (N,3) or null
    const uint32_t *evidx;     // (N,3) or null
    const R *eb0, *eb1, *eweight;
};

template <typename R> struct ScatterArgs {
    int64_t N;
    int K, P;                  // P = number of (N,3) arrays in out_param (5K or 5K-2)
    VertexPtrs<R> v[kMaxVertices];
    ScatterPtrs<R> s[kMaxVertices];
    const R *out_param, *out_light, *out_diffuse;
    int64_t V, B;
};

template <typename R> EPSM_HD bool nz3(V3<R> g) { return g.x != R(0) || g.y != R(0) || g.z != R(0); }
template <typename R> EPSM_HD V3<R> load_out(const R *base, int64_t slot, int64_t N, int64_t i) {
    return load3(base + slot * N * 3, i);
}

template <typename R, typename Sink>
EPSM_HD void scatter_path(const ScatterArgs<R> &A, int64_t i, Sink &sink) {
    for (int k = 1; k <= A.K; ++k) {
        const int it = k - 1;
        const VertexPtrs<R> &v = A.v[it];
        const ScatterPtrs<R> &s = A.s[it];
        const uint32_t mode = s.mode[i];
        uint32_t vi[3] = {s.vidx[3 * i + 0], s.vidx[3 * i + 1], s.vidx[3 * i + 2]};
        const bool idx_ok = vi[0] < (uint64_t) A.V && vi[1] < (uint64_t) A.V && vi[2] < (uint64_t) A.V;
        const bool pos_ok = idx_ok && (mode & kModePos);
        const bool nrm_ok = idx_ok && (mode & kModeNrm);
        const R b0 = v.b0[i], b1 = v.b1[i], b2 = R(1) - b0 - b1;
        const bool has_nm = 5 * it + 4 < A.P;     // epsm.py:559,644: `iteration*5+4 < len(path_grad)`

        // (1) si.p_j * path_grad[5it+j]                                   epsm.py:559-560
        if (pos_ok && has_nm) {
            for (int j = 0; j < 3; ++j) {
                const V3<R> g = load_out(A.out_param, 5 * it + j, A.N, i);
                if (nz3(g)) sink.pos(vi[j], g);
            }
        }
        // (2) si_follow.p * diffuse_grad[it]   (barycentrics detached)     epsm.py:561-562
        if (pos_ok) {
            const V3<R> g = load_out(A.out_diffuse, it, A.N, i);
            if (nz3(g)) { sink.pos(vi[0], g * b0); sink.pos(vi[1], g * b1); sink.pos(vi[2], g * b2); }
        }
        // (3) si_follow.sh_frame.n * path_grad[5it+3] + bsdf_sample.hf * path_grad[5it+4]   epsm.py:644-645
        if (has_nm) {
            const V3<R> gn = load_out(A.out_param, 5 * it + 3, A.N, i);
            if (nz3(gn)) {
                const R sgn = (mode & kModeFlip) ? R(-1) : R(1);
                if (mode & kModeVertexNormals) {
                    if (nrm_ok) {
                        // logged normals are post-flip (mesh.cpp:820-827): n' = sgn * buffer value;
                        // sh = normalize(sum b_j n'_j); d(sh.g)/d n_j = sgn b_j (g - sh (sh.g)) / |n'|
                        const V3<R> n = load3(v.n0, i) * b0 + load3(v.n1, i) * b1 + load3(v.n2, i) * b2;
                        const R il = rsqrt_(dot(n, n));
                        const V3<R> sh = n * il;
                        const V3<R> pg = (gn - sh * dot(sh, gn)) * (il * sgn);
                        sink.nrm(vi[0], pg * b0); sink.nrm(vi[1], pg * b1); sink.nrm(vi[2], pg * b2);
                    }
                } else if (pos_ok) {
                    // flat mesh: sh = sgn * normalize(cross(p1-p0, p2-p0))  (mesh.cpp:729,811)
                    const V3<R> p0 = load3(v.p0, i), p1 = load3(v.p1, i), p2 = load3(v.p2, i);
                    const V3<R> d0 = p1 - p0, d1 = p2 - p0;
                    const V3<R> c = cross(d0, d1);
                    const R il = rsqrt_(dot(c, c));
                    const V3<R> ch = c * il;
                    const V3<R> cb = (gn - ch * dot(ch, gn)) * (il * sgn);
                    const V3<R> d0b = cross(d1, cb), d1b = cross(cb, d0);
                    sink.pos(vi[1], d0b); sink.pos(vi[2], d1b); sink.pos(vi[0], -(d0b + d1b));
                }
            }
            if (s.bsdf_id && s.dhf_dalpha) {
                const uint32_t bid = s.bsdf_id[i];
                if (bid < (uint64_t) A.B) {
                    const V3<R> gm = load_out(A.out_param, 5 * it + 4, A.N, i);
                    if (nz3(gm)) sink.alpha(bid, dot(gm, load3(s.dhf_dalpha, i)));
                }
            }
        }
        // (4) si_direct.p * light_grad[it] * sum(Lr_dir)                    epsm.py:622-627
        if (s.evidx) {
            const uint32_t e0 = s.evidx[3 * i + 0], e1 = s.evidx[3 * i + 1], e2 = s.evidx[3 * i + 2];
            if (e0 < (uint64_t) A.V && e1 < (uint64_t) A.V && e2 < (uint64_t) A.V) {
                const V3<R> g = load_out(A.out_light, it, A.N, i) * s.eweight[i];
                if (nz3(g)) {
                    const R c0 = s.eb0[i], c1 = s.eb1[i];
                    sink.pos(e0, g * c0); sink.pos(e1, g * c1); sink.pos(e2, g * (R(1) - c0 - c1));
                }
            }
        }
    }
}

}  // namespace epsm
