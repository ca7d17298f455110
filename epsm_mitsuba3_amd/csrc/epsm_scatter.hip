// epsm_scatter.hip -- parameter-gradient scatter (include/epsm.h: epsm_scatter).
//
// One lane per path; contributions that are exactly zero (masked paths) issue
// no atomic.  Adds are `global_atomic_add_f32` (no CAS loop on gfx950); their
// order is not fixed, so sums differ from run to run in the last bits -- tests
// compare against a deterministic fp64 sum with a relative tolerance.
#include <string.h>

#include "epsm_common.h"
#include "epsm_scatter_core.h"

using namespace epsm;
using epsm_host::fail;

namespace {

struct AtomicSink {
    float *gpos, *gnrm, *galpha;
    __device__ __forceinline__ void pos(uint32_t v, V3<float> g) const {
        float *p = gpos + 3 * (int64_t) v;
        atomicAdd(p + 0, g.x); atomicAdd(p + 1, g.y); atomicAdd(p + 2, g.z);
    }
    __device__ __forceinline__ void nrm(uint32_t v, V3<float> g) const {
        float *p = gnrm + 3 * (int64_t) v;
        atomicAdd(p + 0, g.x); atomicAdd(p + 1, g.y); atomicAdd(p + 2, g.z);
    }
    __device__ __forceinline__ void alpha(uint32_t b, float g) const { atomicAdd(galpha + b, g); }
};

__global__ __launch_bounds__(256) void epsm_scatter_kernel(ScatterArgs<float> A, AtomicSink sink) {
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= A.N) return;
    scatter_path<float, AtomicSink>(A, i, sink);
}

}  // namespace

extern "C" int epsm_scatter(int variant, int64_t N, int K,
                            const EpsmVertexRecord *verts, const EpsmScatterRecord *sc,
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
    if (V < 0 || B < 0 || V >= 0xFFFFFFFFLL) return fail(EPSM_EINVAL, "epsm_scatter: bad buffer sizes");
    ScatterArgs<float> A;
    memset(&A, 0, sizeof(A));
    A.N = N; A.K = K; A.P = epsm_num_param_grads(variant, K);
    A.out_param = out_param; A.out_light = out_light; A.out_diffuse = out_diffuse;
    A.V = V; A.B = grad_alpha ? B : 0;
    for (int k = 0; k < K; ++k) {
        const EpsmVertexRecord &v = verts[k];
        const EpsmScatterRecord &s = sc[k];
        if (!v.p0 || !v.p1 || !v.p2 || !v.n0 || !v.n1 || !v.n2 || !v.b0 || !v.b1 || !s.vidx || !s.mode)
            return fail(EPSM_EINVAL, "epsm_scatter: NULL pointer in a vertex / scatter record");
        if (s.evidx && (!s.eb0 || !s.eb1 || !s.eweight))
            return fail(EPSM_EINVAL, "epsm_scatter: evidx given without eb0/eb1/eweight");
        VertexPtrs<float> &o = A.v[k];
        o.p0 = (const float *) v.p0; o.p1 = (const float *) v.p1; o.p2 = (const float *) v.p2;
        o.n0 = (const float *) v.n0; o.n1 = (const float *) v.n1; o.n2 = (const float *) v.n2;
        o.b0 = (const float *) v.b0; o.b1 = (const float *) v.b1;
        ScatterPtrs<float> &t = A.s[k];
        t.vidx = s.vidx; t.mode = s.mode; t.bsdf_id = s.bsdf_id; t.dhf_dalpha = s.dhf_dalpha;
        t.evidx = s.evidx; t.eb0 = s.eb0; t.eb1 = s.eb1; t.eweight = s.eweight;
    }
    AtomicSink sink{grad_pos, grad_nrm, grad_alpha};
    if (!grad_nrm) return fail(EPSM_EINVAL, "epsm_scatter: grad_nrm is NULL (pass a (V,3) buffer; it stays zero "
                                            "when no mesh has EPSM_MODE_NRM_ATTACHED)");
    hipLaunchKernelGGL(epsm_scatter_kernel, dim3((unsigned) ((N + 255) / 256)), dim3(256), 0,
                       (hipStream_t) stream, A, sink);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return epsm_host::hip_fail("epsm_scatter", e);
    return EPSM_OK;
}
