// epsm_trace_reparam.hip -- kernel + C ABI of the reparameterised backward pass (include/epsm_trace.h,
// epsm_trace_paths_reparam; per-path code: epsm_trace_reparam.h).
#include <stdio.h>
#include <string.h>

#include "epsm_common.h"
#include "epsm_trace_reparam.h"
#include "epsm_trace_packet.h"

using namespace epsm;
using epsm_host::fail;

namespace {

#ifndef EPSM_RP_THREADS
#define EPSM_RP_THREADS 64
#endif
#ifndef EPSM_RP_OCC
#define EPSM_RP_OCC 2                   // 256 registers: 3.4 KB of scratch per lane instead of 4.0 at four waves; render_backward 47.4 -> 44.9 ms (3: 46.5, 1: 45.2)
#endif
// Stage 1: one lane = one path, replayed; every vertex differentiated (dual numbers, scratch: three vertex records);
// its warps are left as requests.  (First version, auxiliary rays traced by the same lane: 182 ms per render_backward
// at 4.26 M paths / 128 k triangles / 16 rays with one wave per SIMD, 99 ms with four.)
__global__ __launch_bounds__(EPSM_RP_THREADS, EPSM_RP_OCC) void epsm_reparam_path_kernel(rp::ReparamArgs R, rp::WarpReq *req, int *count) {
    constexpr int kLds = 32;
    __shared__ uint32_t s_stack[kLds * EPSM_RP_THREADS];
    uint32_t deep[kBvhStack - kLds];
    const int64_t i = (int64_t) blockIdx.x * EPSM_RP_THREADS + threadIdx.x;
    BvhStack st{s_stack + threadIdx.x, EPSM_RP_THREADS};
    st.cap = kLds; st.ovf = deep; st.ovf_stride = 1;
    if (i >= R.A.N) return;
    rp::QueueSink sink{req, R.A.N, i, 0};
    // (the camera rays of a wave walked together first, as the tracers do: 41.6 -> 42.3 ms per call; not kept)
    rp::reparam_one_path(R, i, st, sink);
    count[i] = sink.n;
}

// Stage 2: one lane = one auxiliary ray, a group of G lanes (G = 16, 32 or 64 >= reparam_rays) = one request.  The rays of a wave
// walk the tree TOGETHER (epsm_trace_packet.h; round 5): 46.3 -> 42.5 ms per call at 4.26 M paths / 16 rays, 26.5 -> 22.0 ms at
// 1.08 M paths / 64 rays (-DEPSM_RP_NO_WARP_PACKET: every lane its own walk).  A workgroup
// serves the requests of 256 consecutive paths: it lists the ones that exist -- (call n, path), n-major, so that neighbouring
// groups hold the same call of neighbouring paths: rays that start next to each other and point the same way -- and works
// through the list 256 / G requests at a time.  (Round 3 launched one group per (n, path) slot and let the empty ones leave:
// 29 % of the slots exist -- 2.06 requests per path of 7 at max_depth 3 -- and the waves that held any were 75 % full.)
// Z, dZ and the origin's adjoint are reduced over the group with xor shuffles.
template <int G>
__global__ __launch_bounds__(256) void epsm_reparam_warp_kernel(rp::ReparamArgs R, const rp::WarpReq *req, const int *count, int n_max) {
    constexpr int kLds = 32, kPaths = 256, kGroups = 256 / G;
#ifndef EPSM_RP_NO_WARP_PACKET
    __shared__ uint32_t s_pstack[kPacketStack * 4];                        // one column per wave (epsm_trace_packet.h)
#else
    __shared__ uint32_t s_stack[kLds * 256];
#endif
    __shared__ uint16_t s_list[kPaths * rp::kMaxReq];                      // (n << 8) | path of the block
    __shared__ int s_off[rp::kMaxReq * 4 + 1];
    uint32_t deep[kBvhStack - kLds];
    const int64_t N = R.A.N, p0 = (int64_t) blockIdx.x * kPaths;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    int c = p0 + tid < N ? count[p0 + tid] : 0;
    c = c < n_max ? c : n_max;
    int rank[rp::kMaxReq];
#pragma unroll
    for (int n = 0; n < rp::kMaxReq; ++n) {
        const unsigned long long m = __ballot(c > n);
        rank[n] = __builtin_amdgcn_mbcnt_hi((unsigned) (m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) m, 0u));
        if (lane == 0) s_off[n * 4 + wv] = __popcll(m);
    }
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int e = 0; e < rp::kMaxReq * 4; ++e) { const int v = s_off[e]; s_off[e] = run; run += v; }
        s_off[rp::kMaxReq * 4] = run;
    }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < rp::kMaxReq; ++n)
        if (c > n) s_list[s_off[n * 4 + wv] + rank[n]] = (uint16_t) ((n << 8) | tid);
    __syncthreads();
    const int total = s_off[rp::kMaxReq * 4];
    const int r = tid % G;
#ifndef EPSM_RP_NO_WARP_PACKET
    (void) deep; (void) kLds;
    for (int b0 = 0; b0 < total; b0 += kGroups) {                          // (uniform over the workgroup: the wave walks its rays together)
        const int b = b0 + tid / G;
        const bool live = b < total;
        const int e = s_list[live ? b : 0], n = e >> 8;
#else
    constexpr bool live = true;
    BvhStack st{s_stack + threadIdx.x, 256};
    st.cap = kLds; st.ovf = deep; st.ovf_stride = 1;
    for (int b = tid / G; b < total; b += kGroups) {                       // (uniform over the group)
        const int e = s_list[b], n = e >> 8;
#endif
        const int64_t i = p0 + (e & 255);
        const rp::WarpReq q = req[(int64_t) n * N + i];
        const F3 o = f3(q.o[0], q.o[1], q.o[2]), d = f3(q.d[0], q.d[1], q.d[2]), g_dir = f3(q.gdir[0], q.gdir[1], q.gdir[2]);
        F3 fs, ft;
        coordinate_system(d, fs, ft);
        rp::Aux A;
        A.w = 0.f; A.dw = zero3<float>(); A.v = d; A.tri = kNoIndex; A.b1 = A.b2 = A.inv_dist = 0.f;
        const bool mine = live && r < R.cfg.rays;
#ifndef EPSM_RP_NO_WARP_PACKET
        // the rays of a request leave one point within a fraction of a degree, the requests of a wave are the same call of
        // neighbouring paths: the wave walks the tree once for all of them
        rp::AuxDraw D; D.ray.o = o; D.ray.d = d; D.ray.maxt = 0.f; D.tangent = zero3<float>(); D.sy_ = 0.f;
        if (mine) D = rp::aux_begin(R.cfg, rp::WarpId{0xffffffffu ^ R.A.seed, (uint32_t) (R.A.path_offset + i), n}, r, o, d, fs, ft);
        const TriHit ath = packet_intersect<false>(R.A.S, D.ray, mine, s_pstack + wv * kPacketStack);
        if (mine) A = rp::aux_finish(R.A.S, R.cfg, D, ath, o, d);
#else
        if (mine) A = rp::aux_ray(R.A.S, R.cfg, rp::WarpId{0xffffffffu ^ R.A.seed, (uint32_t) (R.A.path_offset + i), n}, r, o, d, fs, ft, st);
#endif
        float Z = A.w; F3 dZ = A.dw;
#pragma unroll
        for (int m = 1; m < G; m <<= 1) { Z += __shfl_xor(Z, m); dZ.x += __shfl_xor(dZ.x, m); dZ.y += __shfl_xor(dZ.y, m); dZ.z += __shfl_xor(dZ.z, m); }
        // the adjoint of reparam.py:269-327 at V = 0 (warp_backward, one auxiliary ray per lane)
        Z = fmaxf(Z, 1e-8f);
        const float iZ = 1.f / Z;
        const F3 g_V = (g_dir - d * dot(d, g_dir)) * iZ - dZ * (q.gdiv * iZ * iZ);
        const F3 g_v = mine ? g_V * A.w + A.dw * (q.gdiv * iZ) : zero3<float>();
        F3 g_o = zero3<float>(), g_d = zero3<float>(), g_p = zero3<float>();
        uint32_t pending = 0xFFFFFFFFu;                                    // the triangle this lane still owes its share to
        if (mine) {
            if (A.tri == kNoIndex) g_d = g_v;
            else {
                g_p = (g_v - A.v * dot(A.v, g_v)) * A.inv_dist;
                g_o = -g_p;
                if (R.A.S.meshes[R.A.S.tri_mesh[A.tri]].flags & EPSM_MESH_POS_ATTACHED) pending = A.tri;
            }
        }
        // The rays of a warp mostly hit the same one or two triangles: their shares are summed over the group first, triangle
        // by triangle, and added by one lane.  (One float atomic per ray loses the small ones: a vertex's sum over 10^7 rays
        // is 10^5 times a single share, which then falls under half an ulp of it -- 2 % of a four-vertex wall's gradient at
        // 256 spp.)
        for (int round = 0; round < G; ++round) {
            uint32_t t = pending;
#pragma unroll
            for (int m = 1; m < G; m <<= 1) { const uint32_t u = (uint32_t) __shfl_xor((int) t, m); t = u < t ? u : t; }
            if (t == 0xFFFFFFFFu) break;                                   // (uniform over the group)
            const bool sel = pending == t;
            const float b1 = sel ? A.b1 : 0.f, b2 = sel ? A.b2 : 0.f, b0 = sel ? 1.f - A.b1 - A.b2 : 0.f;
            float acc[9] = {g_p.x * b0, g_p.y * b0, g_p.z * b0, g_p.x * b1, g_p.y * b1, g_p.z * b1, g_p.x * b2, g_p.y * b2, g_p.z * b2};
#pragma unroll
            for (int m = 1; m < G; m <<= 1)
#pragma unroll
                for (int k = 0; k < 9; ++k) acc[k] += __shfl_xor(acc[k], m);
            if (r == 0) {
                const uint32_t *iv = R.A.S.tri + 3 * (int64_t) t;
                rp::add_vertex(R.G.pos, iv[0], f3(acc[0], acc[1], acc[2])); rp::add_vertex(R.G.pos, iv[1], f3(acc[3], acc[4], acc[5]));
                rp::add_vertex(R.G.pos, iv[2], f3(acc[6], acc[7], acc[8]));
            }
            if (sel) pending = 0xFFFFFFFFu;
        }
#pragma unroll
        for (int m = 1; m < G; m <<= 1) {
            g_o.x += __shfl_xor(g_o.x, m); g_o.y += __shfl_xor(g_o.y, m); g_o.z += __shfl_xor(g_o.z, m);
            g_d.x += __shfl_xor(g_d.x, m); g_d.y += __shfl_xor(g_d.y, m); g_d.z += __shfl_xor(g_d.z, m);
        }
        if (live && r == 0 && q.ftri != kNoIndex) {
            if (q.em_inv_dist != 0.f) g_o = g_o - (g_d - d * dot(d, g_d)) * q.em_inv_dist;
            rp::add_follow_point(R.A.S, R.G, q.ftri, q.fb1, q.fb2, g_o);
        }
    }
}

size_t req_bytes(int64_t N) { return ((size_t) N * rp::kMaxReq * sizeof(rp::WarpReq) + 255) & ~(size_t) 255; }

}  // namespace

extern "C" size_t epsm_trace_reparam_workspace_bytes(int64_t N) { return N > 0 ? req_bytes(N) + (size_t) N * sizeof(int) : 0; }

extern "C" int epsm_trace_paths_reparam(const EpsmScene *scene, const EpsmSensor *sensor,
                                        uint32_t seed, int spp, int max_depth, int rr_depth,
                                        int64_t path_offset, int64_t N,
                                        const float *radiance, const float *adj_radiance, const float *adj_film,
                                        int reparam_max_depth, int reparam_rays, float kappa, float exponent, uint32_t flags,
                                        float *grad_pos, float *grad_nrm, void *workspace, size_t workspace_bytes, void *stream) {
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
    if (!workspace || (((uintptr_t) workspace) & 15) || workspace_bytes < epsm_trace_reparam_workspace_bytes(N))
        return bad("workspace: 16-byte aligned, >= epsm_trace_reparam_workspace_bytes(N)");
    if (reparam_rays < 1 || reparam_rays > rp::kMaxAux || reparam_max_depth < 0 || !(kappa > 0.f) || !(exponent > 0.f))
        return bad("need 1 <= reparam_rays <= 64, reparam_max_depth >= 0, kappa > 0, exponent > 0");
    if (flags & ~EPSM_REPARAM_ANTITHETIC) return bad("unknown flag");
    if (scene->n_triangles <= 0 || !scene->positions || !scene->normals || !scene->tri || !scene->tri_mesh || !scene->meshes ||
        !scene->bsdfs || !scene->bvh || !scene->prim_index || !scene->tri_verts)
        return bad("NULL scene array");
    if (const char *why = epsm_host::scene_tables_invalid(scene)) return bad(why);
    rp::ReparamArgs R;
    memset(&R, 0, sizeof(R));
    R.A.S = *scene; R.A.C = *sensor;
    R.A.seed = seed; R.A.spp = spp; R.A.max_depth = max_depth; R.A.rr_depth = rr_depth; R.A.K_log = 0;
    R.A.path_offset = path_offset; R.A.N = N;
    R.cfg.max_depth = reparam_max_depth; R.cfg.rays = reparam_rays; R.cfg.kappa = kappa; R.cfg.exponent = exponent; R.cfg.flags = flags;
    R.radiance = radiance; R.adj_radiance = adj_radiance; R.adj_film = adj_film;
    R.G.pos = grad_pos; R.G.nrm = grad_nrm;
    rp::WarpReq *req = (rp::WarpReq *) workspace;
    int *count = (int *) ((char *) workspace + req_bytes(N));
    hipLaunchKernelGGL(epsm_reparam_path_kernel, dim3((unsigned) ((N + EPSM_RP_THREADS - 1) / EPSM_RP_THREADS)), dim3(EPSM_RP_THREADS), 0,
                       (hipStream_t) stream, R, req, count);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return epsm_host::hip_fail("epsm_trace_paths_reparam", e);
    if (reparam_max_depth > 0) {
        // the n-th requests of all paths, n = 0 .. n_max - 1: the camera ray + two warps per vertex the depths allow
        const int depth = max_depth < 6 ? max_depth : 6;
        const int n_max = 1 + 2 * depth < rp::kMaxReq ? 1 + 2 * depth : rp::kMaxReq;
        const int G = reparam_rays <= 16 ? 16 : reparam_rays <= 32 ? 32 : 64;
        const dim3 grid((unsigned) ((N + 255) / 256));                     // one workgroup per 256 paths (N < 2^32: checked above)
        if (G == 16) hipLaunchKernelGGL(epsm_reparam_warp_kernel<16>, grid, dim3(256), 0, (hipStream_t) stream, R, req, count, n_max);
        else if (G == 32) hipLaunchKernelGGL(epsm_reparam_warp_kernel<32>, grid, dim3(256), 0, (hipStream_t) stream, R, req, count, n_max);
        else hipLaunchKernelGGL(epsm_reparam_warp_kernel<64>, grid, dim3(256), 0, (hipStream_t) stream, R, req, count, n_max);
        e = hipGetLastError();
        if (e != hipSuccess) return epsm_host::hip_fail("epsm_trace_paths_reparam", e);
    }
    return EPSM_OK;
}
