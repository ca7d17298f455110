// epsm_trace_reparam.hip -- kernel + C ABI of the reparameterised backward pass (include/epsm_trace.h,
// epsm_trace_paths_reparam; per-path code: epsm_trace_reparam.h).
#include <stdio.h>
#include <string.h>

#include "epsm_common.h"
#include "epsm_trace_reparam.h"

using namespace epsm;
using epsm_host::fail;

namespace {

// One lane = one path, replayed with all its auxiliary rays.  The three vertex records, the warp's auxiliary-ray table
// (2.8 KB) and the dual numbers live in scratch: this pass is bound by its 16..64 closest-hit traversals per warp, not by
// the bookkeeping around them.  64-thread workgroups: paths of neighbouring samples, whose auxiliary rays stay together.
// Waves per SIMD (4.26 M paths, 128 k triangles, 16 rays, ms per render_backward): 1 (256 + 256 registers, 5.4 KB of
// scratch) 181, 2: 127, 3: 108, 4 (128 registers, 6.7 KB) 100, 5: 103; 128-thread workgroups at 2: 137 against 127.
#ifndef EPSM_RP_THREADS
#define EPSM_RP_THREADS 64
#endif
#ifndef EPSM_RP_OCC
#define EPSM_RP_OCC 4
#endif
__global__ __launch_bounds__(EPSM_RP_THREADS, EPSM_RP_OCC) void epsm_reparam_kernel(rp::ReparamArgs R) {
    constexpr int kLds = 32;
    __shared__ uint32_t s_stack[kLds * EPSM_RP_THREADS];
    uint32_t deep[kBvhStack - kLds];
    const int64_t i = (int64_t) blockIdx.x * EPSM_RP_THREADS + threadIdx.x;
    if (i >= R.A.N) return;
    BvhStack st{s_stack + threadIdx.x, EPSM_RP_THREADS};
    st.cap = kLds; st.ovf = deep; st.ovf_stride = 1;
    rp::Warp W;
    rp::reparam_one_path(R, i, st, W);
}

}  // namespace

extern "C" int epsm_trace_paths_reparam(const EpsmScene *scene, const EpsmSensor *sensor,
                                        uint32_t seed, int spp, int max_depth, int rr_depth,
                                        int64_t path_offset, int64_t N,
                                        const float *radiance, const float *adj_radiance, const float *adj_film,
                                        int reparam_max_depth, int reparam_rays, float kappa, float exponent,
                                        float *grad_pos, float *grad_nrm, void *stream) {
    epsm_host::err_buf()[0] = 0;
    auto bad = [&](const char *what) { char msg[200]; snprintf(msg, sizeof(msg), "epsm_trace_paths_reparam: %s", what); return fail(EPSM_EINVAL, msg); };
    if (!scene || !sensor) return bad("NULL scene / sensor");
    if (N == 0) return EPSM_OK;
    if (N < 0 || spp < 1 || max_depth < 1 || rr_depth < 1 || path_offset < 0) return bad("bad N / spp / max_depth / rr_depth / path_offset");
    if (sensor->border < 0 || sensor->border > 8) return bad("bad sensor border");
    if (path_offset + N > (int64_t) (sensor->width + 2 * sensor->border) * (sensor->height + 2 * sensor->border) * spp ||
        path_offset + N > 0xFFFFFFFFLL)
        return bad("path range exceeds (width + 2 border) * (height + 2 border) * spp (or 2^32)");
    if (!radiance || !adj_radiance || !adj_film || !grad_pos) return bad("NULL per-path input or grad_pos");
    if (reparam_rays < 1 || reparam_rays > rp::kMaxAux || reparam_max_depth < 0 || !(kappa > 0.f) || !(exponent > 0.f))
        return bad("need 1 <= reparam_rays <= 64, reparam_max_depth >= 0, kappa > 0, exponent > 0");
    if (scene->n_triangles <= 0 || !scene->positions || !scene->normals || !scene->tri || !scene->tri_mesh || !scene->meshes ||
        !scene->bsdfs || !scene->bvh || !scene->prim_index || !scene->tri_verts)
        return bad("NULL scene array");
    if (scene->n_emitters > 0 && !scene->emitters) return bad("NULL emitters");
    rp::ReparamArgs R;
    memset(&R, 0, sizeof(R));
    R.A.S = *scene; R.A.C = *sensor;
    R.A.seed = seed; R.A.spp = spp; R.A.max_depth = max_depth; R.A.rr_depth = rr_depth; R.A.K_log = 0;
    R.A.path_offset = path_offset; R.A.N = N;
    R.cfg.max_depth = reparam_max_depth; R.cfg.rays = reparam_rays; R.cfg.kappa = kappa; R.cfg.exponent = exponent;
    R.radiance = radiance; R.adj_radiance = adj_radiance; R.adj_film = adj_film;
    R.G.pos = grad_pos; R.G.nrm = grad_nrm;
    hipLaunchKernelGGL(epsm_reparam_kernel, dim3((unsigned) ((N + EPSM_RP_THREADS - 1) / EPSM_RP_THREADS)), dim3(EPSM_RP_THREADS), 0, (hipStream_t) stream, R);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return epsm_host::hip_fail("epsm_trace_paths_reparam", e);
    return EPSM_OK;
}
