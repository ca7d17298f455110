// epsm_trace.hip -- kernels + C ABI of the wavefront tracer (include/epsm_trace.h).
#include <string.h>

#include "epsm_common.h"
#include "epsm_trace_core.h"

using namespace epsm;
using epsm_host::fail;

namespace {

__global__ __launch_bounds__(128) void epsm_trace_kernel(TraceArgs A) {
    const int64_t i = (int64_t) blockIdx.x * 128 + threadIdx.x;
    if (i >= A.N) return;
    trace_one_path(A, i);
}

// ImageBlock::put (src/render/imageblock.cpp) with the reconstruction filter evaluated
// exactly (box: the pixel under the sample; gaussian: stddev 0.5, radius 4 sigma = 2,
// src/rfilters/gaussian.cpp) -- separable weights, float atomics into [r,g,b,w].
__global__ __launch_bounds__(256) void epsm_film_splat_kernel(int64_t N, const float *pos, const float *rad, int W, int H,
                                                              int rfilter, float *accum) {
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float px = pos[2 * i], py = pos[2 * i + 1];
    const float r = rad[3 * i], g = rad[3 * i + 1], b = rad[3 * i + 2];
    if (rfilter == EPSM_RFILTER_BOX) {
        const int x = (int) floorf(px), y = (int) floorf(py);
        if (x < 0 || y < 0 || x >= W || y >= H) return;
        float *a = accum + 4 * ((int64_t) y * W + x);
        atomicAdd(a + 0, r); atomicAdd(a + 1, g); atomicAdd(a + 2, b); atomicAdd(a + 3, 1.f);
        return;
    }
    const float radius = 2.f, alpha = -1.f / (2.f * 0.5f * 0.5f), bias = expf(alpha * radius * radius);
    const int x0 = (int) ceilf(px - radius - 0.5f), x1 = (int) floorf(px + radius - 0.5f);
    const int y0 = (int) ceilf(py - radius - 0.5f), y1 = (int) floorf(py + radius - 0.5f);
    for (int y = y0; y <= y1; ++y) {
        if (y < 0 || y >= H) continue;
        const float dy = (y + 0.5f) - py, wy = fmaxf(0.f, expf(alpha * dy * dy) - bias);
        for (int x = x0; x <= x1; ++x) {
            if (x < 0 || x >= W) continue;
            const float dx = (x + 0.5f) - px, w = wy * fmaxf(0.f, expf(alpha * dx * dx) - bias);
            if (w == 0.f) continue;
            float *a = accum + 4 * ((int64_t) y * W + x);
            atomicAdd(a + 0, r * w); atomicAdd(a + 1, g * w); atomicAdd(a + 2, b * w); atomicAdd(a + 3, w);
        }
    }
}
__global__ __launch_bounds__(256) void epsm_film_develop_kernel(int64_t n, const float *accum, float *image) {
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float w = accum[4 * i + 3], iw = w != 0.f ? 1.f / w : 0.f;
    image[3 * i] = accum[4 * i] * iw; image[3 * i + 1] = accum[4 * i + 1] * iw; image[3 * i + 2] = accum[4 * i + 2] * iw;
}

}  // namespace

extern "C" int epsm_trace_paths(const EpsmScene *scene, const EpsmSensor *sensor,
                                uint32_t seed, int spp, int max_depth, int rr_depth,
                                int64_t path_offset, int64_t N, int K_log,
                                float *ray_o, float *ray_d, float *ray_dx, float *ray_dy,
                                float *film_pos, float *radiance, uint8_t *valid,
                                const EpsmRecordOut *recs, void *stream) {
    epsm_host::err_buf()[0] = 0;
    if (!scene || !sensor) return fail(EPSM_EINVAL, "epsm_trace_paths: NULL scene / sensor");
    if (N == 0) return EPSM_OK;
    if (N < 0 || spp < 1 || max_depth < 1 || rr_depth < 1 || path_offset < 0)
        return fail(EPSM_EINVAL, "epsm_trace_paths: bad N / spp / max_depth / rr_depth / path_offset");
    if (K_log < 0 || K_log > EPSM_MAX_VERTICES || K_log > (max_depth < 6 ? max_depth : 6))
        return fail(EPSM_EINVAL, "epsm_trace_paths: K_log must be <= min(max_depth, 5)");
    if (path_offset + N > (int64_t) sensor->width * sensor->height * spp || path_offset + N > 0xFFFFFFFFLL)
        return fail(EPSM_EINVAL, "epsm_trace_paths: path range exceeds width*height*spp (or 2^32, common.py:468-475)");
    if (!ray_o || !ray_d || !ray_dx || !ray_dy || (K_log > 0 && !recs))
        return fail(EPSM_EINVAL, "epsm_trace_paths: NULL output");
    if (scene->n_triangles > 0 && (!scene->positions || !scene->normals || !scene->tri || !scene->tri_mesh ||
                                   !scene->meshes || !scene->bsdfs || !scene->bvh || !scene->prim_index))
        return fail(EPSM_EINVAL, "epsm_trace_paths: NULL scene array");
    if (scene->n_emitters > 0 && !scene->emitters) return fail(EPSM_EINVAL, "epsm_trace_paths: NULL emitters");
    TraceArgs A;
    memset(&A, 0, sizeof(A));
    A.S = *scene; A.C = *sensor;
    A.seed = seed; A.spp = spp; A.max_depth = max_depth; A.rr_depth = rr_depth; A.K_log = K_log;
    A.path_offset = path_offset; A.N = N;
    A.ray_o = ray_o; A.ray_d = ray_d; A.ray_dx = ray_dx; A.ray_dy = ray_dy;
    A.film_pos = film_pos; A.radiance = radiance; A.valid = valid;
    for (int k = 0; k < K_log; ++k) {
        const EpsmRecordOut &r = recs[k];
        if (!r.p0 || !r.p1 || !r.p2 || !r.n0 || !r.n1 || !r.n2 || !r.b0 || !r.b1 || !r.eta || !r.hf || !r.light ||
            !r.bsdf || !r.active || !r.active_em || !r.ismesh || !r.tri || !r.aux || !r.emit)
            return fail(EPSM_EINVAL, "epsm_trace_paths: NULL pointer in a record (p / normal may be NULL)");
        A.rec[k] = r;
    }
    hipLaunchKernelGGL(epsm_trace_kernel, dim3((unsigned) ((N + 127) / 128)), dim3(128), 0, (hipStream_t) stream, A);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return epsm_host::hip_fail("epsm_trace_paths", e);
    return EPSM_OK;
}

extern "C" int epsm_film_splat(int64_t N, const float *film_pos, const float *radiance, int width, int height,
                               int rfilter, float *accum, void *stream) {
    epsm_host::err_buf()[0] = 0;
    if (N == 0) return EPSM_OK;
    if (N < 0 || !film_pos || !radiance || !accum || width < 1 || height < 1)
        return fail(EPSM_EINVAL, "epsm_film_splat: bad argument");
    hipLaunchKernelGGL(epsm_film_splat_kernel, dim3((unsigned) ((N + 255) / 256)), dim3(256), 0, (hipStream_t) stream,
                       N, film_pos, radiance, width, height, rfilter, accum);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return epsm_host::hip_fail("epsm_film_splat", e);
    return EPSM_OK;
}
extern "C" int epsm_film_develop(int width, int height, const float *accum, float *image, void *stream) {
    epsm_host::err_buf()[0] = 0;
    if (!accum || !image || width < 1 || height < 1) return fail(EPSM_EINVAL, "epsm_film_develop: bad argument");
    const int64_t n = (int64_t) width * height;
    hipLaunchKernelGGL(epsm_film_develop_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, (hipStream_t) stream,
                       n, accum, image);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return epsm_host::hip_fail("epsm_film_develop", e);
    return EPSM_OK;
}
