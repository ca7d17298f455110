// epsm_scatter.hip -- parameter-gradient scatter (include/epsm.h: epsm_scatter).
//
// One lane per path, vertices in lock step so that the items of a wave line up;
// items are merged across the wave (epsm_wave_scatter.h) before the
// `global_atomic_add_f32`s.  Atomic order is not fixed, so sums differ from run to
// run in the last bits -- tests compare against a deterministic fp64 sum.
#include <stdlib.h>
#include <string.h>

#include "epsm_common.h"
#include "epsm_scatter_core.h"
#include "epsm_wave_scatter.h"

using namespace epsm;
using epsm_host::fail;

namespace {

struct Targets { float *gpos, *gnrm, *galpha; };

constexpr int kScatterBlocks = 2048;            // persistent workgroups (8 per CU by count, 5 resident by LDS)
// (table rows: 2048 = 32 KB.  1024 / 1536 / 3072 / 4096 / 4800 rows measured: 4.6 -> 8.5 / 5.6 / 5.6 / 6.8 / 6.8 ms on
// config 2 -- too few rows, or too few resident workgroups -- and only the specular profile gains from more, 17.8 -> 12 ms)

// One lane per path within a 256-path chunk; a workgroup walks a CONTIGUOUS range of
// chunks (neighbouring pixels -> the same triangles come back -> they stay in its table).
template <int BITS>
__global__ __launch_bounds__(256) void epsm_scatter_kernel(ScatterArgs<float> A, Targets tg, int64_t chunks_per_block) {
    constexpr int kTableSize = BITS;      // rows
    __shared__ uint32_t s_keys[kTableSize];
    __shared__ float s_vals[kTableSize * 3];
    __shared__ int s_used;
    const LdsTable<BITS> T{s_keys, s_vals, &s_used, tg.gpos, tg.gnrm, tg.galpha, (uint32_t) A.V};
    T.clear();
    const int64_t first = (int64_t) blockIdx.x * chunks_per_block;
#pragma unroll 1
    for (int64_t c = first; c < first + chunks_per_block; ++c) {
        const int64_t i0 = c * 256 + threadIdx.x;
        if (c * 256 >= A.N) break;
        const bool in = i0 < A.N;
        const int64_t i = in ? i0 : A.N - 1;      // out-of-range lanes stay in the wave (shuffles) but add nothing
#pragma unroll 1
        for (int it = 0; it < A.K; ++it) {
            const VertexGrads<float> g = load_vertex_grads(A, i, it);
            // skip the vertex for the whole wave when nobody has a gradient there (masked tails)
            const bool live = in && (nz3(g.gp[0]) || nz3(g.gp[1]) || nz3(g.gp[2]) || nz3(g.gn) || nz3(g.gm) ||
                                     nz3(g.glight) || nz3(g.gdiff));
            if (__ballot(live) == 0ull) continue;
            VertexItems<float> q;
            if (live) {
                q = vertex_items(A.v[it], A.s[it], A.tab, i, g, A.V, A.B);
            } else {
                q.pos_ok = q.nrm_ok = q.alpha_ok = q.em_ok = q.sh_ok = false;
                q.si[0] = q.si[1] = q.si[2] = kNoIndex;
                q.vi[0] = q.vi[1] = q.vi[2] = kNoIndex;
                q.ei[0] = q.ei[1] = q.ei[2] = kNoIndex;
                q.bid = kNoIndex; q.alpha = 0.f;
                for (int j = 0; j < 3; ++j) q.pos[j] = q.nrm[j] = q.em[j] = q.sh[j] = zero3<float>();
            }
            const bool pos_v = live && q.pos_ok;
            const bool nrm_v = live && q.nrm_ok && (nz3(q.nrm[0]) || nz3(q.nrm[1]) || nz3(q.nrm[2]));
            // rows of the hit triangle: segmented scan when the wave has <= 16 runs, else straight into the table
            scatter_triangle_adaptive(T, 0u, pos_v, q.vi, q.pos, 16);
            scatter_triangle_adaptive(T, (uint32_t) A.V, nrm_v, q.vi, q.nrm, 16);
            if (tg.galpha && live && q.alpha_ok) T.add(2u * (uint32_t) A.V + q.bid, q.alpha, 0.f, 0.f);
            if (A.s[it].emit) scatter_triangle_direct(T, 0u, live && q.em_ok, q.ei, q.em);
            if (A.s[it].shadow) scatter_triangle_adaptive(T, 0u, live && q.sh_ok, q.si, q.sh, 16);
        }
        if (T.crowded()) T.flush();               // workgroup-uniform census
    }
    T.flush();
}

}  // namespace

extern "C" int epsm_scatter(int variant, int64_t N, int K,
                            const EpsmVertexRecord *verts, const EpsmScatterRecord *sc,
                            const uint32_t *tri_table, int64_t T,
                            const float *out_param, const float *out_light, const float *out_diffuse,
                            float *grad_pos, float *grad_nrm, float *grad_alpha,
                            int64_t V, int64_t B, void *stream) {
    epsm_host::err_buf()[0] = 0;
    if (variant != EPSM_VARIANT_MANIFOLD && variant != EPSM_VARIANT_MANIFOLD_CAUSTIC)
        return fail(EPSM_EINVAL, "epsm_scatter: unknown variant");
    if (K < 1 || K > EPSM_MAX_VERTICES) return fail(EPSM_EINVAL, "epsm_scatter: K must be in 1..5");
    if (N == 0) return EPSM_OK;
    if (N < 0 || (N + 255) / 256 > 0x7fffffffLL) return fail(EPSM_EINVAL, "epsm_scatter: bad N");
    if (!verts || !sc || !out_param || !out_light || !out_diffuse || !grad_pos)
        return fail(EPSM_EINVAL, "epsm_scatter: NULL argument");
    if (T < 0 || (T > 0 && !tri_table) || (((uintptr_t) tri_table) & 15))
        return fail(EPSM_EINVAL, "epsm_scatter: bad triangle table (need T >= 0 rows of 16 B, 16-byte aligned)");
    if (!grad_nrm) return fail(EPSM_EINVAL, "epsm_scatter: grad_nrm is NULL (pass a (V,3) buffer; it stays zero "
                                            "when no mesh has EPSM_MODE_NRM_ATTACHED)");
    if (V < 0 || B < 0 || 2 * V + B >= 0xFFFFFFFFLL) return fail(EPSM_EINVAL, "epsm_scatter: bad buffer sizes (need 2V+B < 2^32-1)");
    ScatterArgs<float> A;
    memset(&A, 0, sizeof(A));
    A.N = N; A.K = K; A.P = epsm_num_param_grads(variant, K);
    A.out_param = out_param; A.out_light = out_light; A.out_diffuse = out_diffuse;
    A.V = V; A.B = grad_alpha ? B : 0;
    A.tab = TriTable{tri_table, T};
    for (int k = 0; k < K; ++k) {
        const EpsmVertexRecord &v = verts[k];
        const EpsmScatterRecord &s = sc[k];
        if (!v.p0 || !v.p1 || !v.p2 || !v.n0 || !v.n1 || !v.n2 || !v.b0 || !v.b1 || !s.tri)
            return fail(EPSM_EINVAL, "epsm_scatter: NULL pointer in a vertex / scatter record");
        if ((((uintptr_t) s.aux) | ((uintptr_t) s.emit) | ((uintptr_t) s.shadow)) & 15 || (((uintptr_t) s.tri) & 3))
            return fail(EPSM_EINVAL, "epsm_scatter: aux/emit/shadow must be 16-byte aligned (tri: 4)");
        VertexPtrs<float> &o = A.v[k];
        o.p0 = (const float *) v.p0; o.p1 = (const float *) v.p1; o.p2 = (const float *) v.p2;
        o.n0 = (const float *) v.n0; o.n1 = (const float *) v.n1; o.n2 = (const float *) v.n2;
        o.b0 = (const float *) v.b0; o.b1 = (const float *) v.b1;
        ScatterPtrs<float> &t = A.s[k];
        t.tri = s.tri; t.aux = s.aux; t.emit = s.emit;
        t.shadow = k == 0 ? s.shadow : nullptr;       // epsm.py:610: `iteration == 0`
    }
    Targets tg{grad_pos, grad_nrm, grad_alpha};
    const int64_t chunks = (N + 255) / 256;
    const int64_t blocks = chunks < kScatterBlocks ? chunks : kScatterBlocks;
    const int64_t chunks_per_block = (chunks + blocks - 1) / blocks;
    // Adaptive run merge (<= 16 runs per wave), 2048-row table (4 workgroups per CU): the winner of the A/B on
    // config 2 over {runs + hot-key rounds, direct LDS atomics, adaptive 8 / 16} x {1024, 2048, 4096 rows}.
    hipLaunchKernelGGL((epsm_scatter_kernel<2048>), dim3((unsigned) blocks), dim3(256), 0, (hipStream_t) stream, A, tg, chunks_per_block);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return epsm_host::hip_fail("epsm_scatter", e);
    return EPSM_OK;
}
