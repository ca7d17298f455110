// epsm_tangent.hip -- first-vertex tangent (include/epsm.h: epsm_first_vertex_tangent).
//
// Closed form of the forward-mode AD block of render_backward (epsm.py:250-272):
// the directional derivative of the Moeller-Trumbore barycentrics
// (include/mitsuba/render/mesh.h:343-365) along the image-space motion
//     grad_d = (d_x - d) gx + (d_y - d) gy,
// mapped to (b0,b1) and the hit point as src/render/mesh.cpp:698-709 does.
// One lane per path, every input a coalesced (N,3) stream; HBM-bound:
// 4*12 (rays) + 3*12 (triangle) + 1 (mask) in, 12 + 4*stride out per path.
#include "epsm_common.h"
#include "epsm_path_core.h"
#include "epsm_tangent_core.h"

using namespace epsm;
using epsm_host::fail;

namespace {

struct TangentArgs {
    int64_t N, path_offset;
    int spp, res, img_width, img_channels;
    const float *o, *d, *dx, *dy, *grad_img, *p0, *p1, *p2;
    const uint8_t *active;
    float *dlduv;
    int64_t dlduv_stride;
    float *dldp;
    float *grad_o_sum;
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

__device__ __forceinline__ void tangent_one(const TangentArgs &A, int64_t i, V3<float> &gd_acc) {
    const TangentIn in{A.path_offset, A.spp, A.res, A.img_width, A.img_channels, A.o, A.d, A.dx, A.dy, A.grad_img};
    const Tangent t = first_vertex_tangent(in, i, A.p0, A.p1, A.p2, A.active[i] != 0);
    gd_acc = gd_acc + t.gd;
    float *row = A.dlduv + i * A.dlduv_stride;
    row[0] = t.db0;
    row[1] = t.db1;
    for (int64_t c = 2; c < A.dlduv_stride; ++c) row[c] = 0.f;
    float *q = A.dldp + 3 * i;
    q[0] = t.dp.x; q[1] = t.dp.y; q[2] = t.dp.z;
}

// Grid-stride over the paths (at most kMaxBlocks workgroups): the camera-origin
// gradient is a sum over ALL paths into three floats, and same-address float atomics
// serialise at the memory side (~25 ns each) -- one atomic triple per workgroup, not
// per wave, keeps that off the critical path (10 ms -> negligible on 16.8 M paths).
constexpr int kMaxBlocks = 4096;

__global__ __launch_bounds__(256) void epsm_tangent_kernel(TangentArgs A) {
    V3<float> acc = zero3<float>();
    const int64_t stride = (int64_t) gridDim.x * 256;
    for (int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x; i < A.N; i += stride) tangent_one(A, i, acc);
    if (A.grad_o_sum) {                                            // epsm.py:260-261: d/d ray.o = -grad_d
        __shared__ float part[4][3];
        const float sx = wave_sum(-acc.x), sy = wave_sum(-acc.y), sz = wave_sum(-acc.z);
        const int w = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { part[w][0] = sx; part[w][1] = sy; part[w][2] = sz; }
        __syncthreads();
        if (threadIdx.x < 3) {
            const float t = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
            atomicAdd(A.grad_o_sum + threadIdx.x, t);
        }
    }
}

}  // namespace

extern "C" int epsm_first_vertex_tangent(int64_t N, int64_t path_offset, int spp, int res,
                                         const float *ray_o, const float *ray_d,
                                         const float *ray_dx, const float *ray_dy,
                                         const float *grad_img, int img_width, int img_channels,
                                         const float *p0, const float *p1, const float *p2,
                                         const uint8_t *active,
                                         float *dlduv, int64_t dlduv_stride, float *dldp,
                                         float *grad_o_sum, void *stream) {
    epsm_host::err_buf()[0] = 0;
    if (N == 0) return EPSM_OK;
    if (N < 0 || (N + 255) / 256 > 0x7fffffffLL) return fail(EPSM_EINVAL, "epsm_first_vertex_tangent: bad N");
    if (spp < 1 || res < 1 || img_width < res || img_channels < 5)
        return fail(EPSM_EINVAL, "epsm_first_vertex_tangent: need spp>=1, res>=1, img_width>=res, img_channels>=5");
    if (path_offset < 0 || (int64_t) res * res * spp < path_offset + N)
        return fail(EPSM_EINVAL, "epsm_first_vertex_tangent: path_offset + N exceeds res*res*spp");
    if (!ray_o || !ray_d || !ray_dx || !ray_dy || !grad_img || !p0 || !p1 || !p2 || !active || !dlduv || !dldp)
        return fail(EPSM_EINVAL, "epsm_first_vertex_tangent: NULL argument");
    if (dlduv_stride < 2) return fail(EPSM_EINVAL, "epsm_first_vertex_tangent: dlduv_stride < 2");
    TangentArgs A{N, path_offset, spp, res, img_width, img_channels, ray_o, ray_d, ray_dx, ray_dy, grad_img, p0, p1, p2,
                  active, dlduv, dlduv_stride, dldp, grad_o_sum};
    const int64_t blocks = (N + 255) / 256 < kMaxBlocks ? (N + 255) / 256 : kMaxBlocks;
    hipLaunchKernelGGL(epsm_tangent_kernel, dim3((unsigned) blocks), dim3(256), 0,
                       (hipStream_t) stream, A);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return epsm_host::hip_fail("epsm_first_vertex_tangent", e);
    return EPSM_OK;
}
