// epsm_grad_scatter.hip -- the ACCUMULATING entry points of the C ABI (include/epsm.h): epsm_manifold_grad_scatter (calc_grad +
// scatter in one launch), epsm_backward_pass (first-vertex tangent + calc_grad + scatter), epsm_backward_pass_packed (the same
// on the native log), epsm_release_workspace -- argument checks, the replica workspace of small wavefronts and its reduction
// kernel.  The kernel behind all three is csrc/epsm_backward_cp.hip (one lane per (path, constraint vertex), round 3); round
// 2's one-lane-per-path fused kernel, which lived here, is gone (profiles/r03_a_knockouts.txt holds the last A/B numbers).
#include <stdlib.h>
#include <atomic>
#include <mutex>
#include <stdio.h>
#include <string.h>

#include "epsm_fused.h"

using namespace epsm;
using epsm_host::fail;

namespace epsm {

// Sums the replicas of a small wavefront into the caller's buffers and leaves the workspace zero for the next launch.
__global__ __launch_bounds__(256) void reduce_replicas_kernel(float *rep, int replicas, int64_t stride, int64_t V, int64_t B,
                                                              float *gpos, float *gnrm, float *galpha, float *go) {
    const int64_t e = (int64_t) blockIdx.x * 256 + threadIdx.x, n = 6 * V + B + 3;
    if (e >= n) return;
    // all loads first (a load behind a store to the same array waits for it: 32 round trips in a row, 9.9 us for config 5)
    float sum = 0.f;
#pragma unroll 8
    for (int r = 0; r < replicas; ++r) sum += __builtin_nontemporal_load(rep + r * stride + e);
    for (int r = 0; r < replicas; ++r) rep[r * stride + e] = 0.f;
    float *dst = e < 3 * V ? gpos + e : e < 6 * V ? gnrm + (e - 3 * V) : e < 6 * V + B ? (galpha ? galpha + (e - 6 * V) : nullptr)
                                                                                      : (go ? go + (e - 6 * V - B) : nullptr);
    if (dst && sum != 0.f && fabsf(sum) < __builtin_inff()) atomicAdd(dst, sum);   // (a non-finite sum adds nothing)      // atomic like every other add into the caller's buffers: launches on other streams may share them
}

// Why replicas: the workgroups of a small wavefront all flush at the same moment, and every one of them holds the rows
// every path adds to (the emitter's vertices, the alpha slots, the camera origin).  Global float atomics on ONE 12-byte row
// retire at 12.7 ns each, on 96 rows at 0.86 G rows/s, on rows drawn from >= 15 000 at 18.6 G rows/s
// (tools/micro/global_atomics.hip): 512 workgroups x 96 emitter rows = 57 us of a 190 us kernel (config 5), 2048
// workgroups four times that.  With the rows of workgroup b in replica b % R the same atomics spread over R x as many
// rows; one small kernel sums the replicas afterwards.  The workspace is the library's, one per (device, stream), kept
// zero between launches.
struct Workspace { int dev; hipStream_t stream; float *p; size_t bytes; };
static Workspace g_ws[16];
static int g_ws_n = 0;
static std::mutex g_ws_mutex;
hipError_t fused_workspace(hipStream_t s, size_t bytes, float **out) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    Workspace *w = nullptr;
    for (int i = 0; i < g_ws_n; ++i) if (g_ws[i].dev == dev && g_ws[i].stream == s) w = &g_ws[i];
    if (!w) {
        if (g_ws_n == 16) { *out = nullptr; return hipSuccess; }          // no replicas for a 17th stream
        w = &g_ws[g_ws_n++];
        *w = Workspace{dev, s, nullptr, 0};
    }
    if (w->bytes < bytes) {
        if (w->p) { e = hipFree(w->p); w->p = nullptr; w->bytes = 0; if (e != hipSuccess) return e; }
        e = hipMalloc((void **) &w->p, kReplicaBudget + 4096);
        if (e != hipSuccess) { w->p = nullptr; return e; }
        w->bytes = kReplicaBudget;
        e = hipMemsetAsync(w->p, 0, kReplicaBudget + 4096, s);
        if (e != hipSuccess) return e;
    }
    *out = w->p;
    return hipSuccess;
}

void fused_workspace_invalidate(hipStream_t s) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    for (int i = 0; i < g_ws_n; ++i)
        if (g_ws[i].dev == dev && g_ws[i].stream == s) g_ws[i].bytes = 0;          // fused_workspace(): free, allocate, memset
}

hipError_t fused_release_workspaces() {
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    hipError_t first = hipSuccess;
    int here = 0;
    (void) hipGetDevice(&here);
    for (int i = 0; i < g_ws_n; ++i) {
        if (!g_ws[i].p) continue;
        hipError_t e = hipSetDevice(g_ws[i].dev);
        if (e == hipSuccess) e = hipFree(g_ws[i].p);
        if (e != hipSuccess && first == hipSuccess) first = e;
    }
    g_ws_n = 0;
    (void) hipSetDevice(here);
    return first;
}

}  // namespace epsm

// Shared by the two entry points: validates the records, fills FusedArgs.  Returns EPSM_OK or fails with `who` in the text.
static int fill_args(FusedArgs &F, const char *who, int variant, int64_t N, int K, const float *cam,
                     const EpsmVertexRecord *verts, const EpsmScatterRecord *sc, const uint32_t *tri_table, int64_t T, float clip,
                     float *grad_pos, float *grad_nrm, float *grad_alpha, int64_t V, int64_t B) {
    char msg[160];
    auto bad = [&](const char *what) { snprintf(msg, sizeof(msg), "%s: %s", who, what); return fail(EPSM_EINVAL, msg); };
    if (variant != EPSM_VARIANT_MANIFOLD && variant != EPSM_VARIANT_MANIFOLD_CAUSTIC) return bad("unknown variant");
    if (K < 1 || K > EPSM_MAX_VERTICES) return bad("K must be in 1..5");
    if (N < 0) return bad("bad N");
    if (!cam || !verts || !sc || !grad_pos || !grad_nrm) return bad("NULL argument");
    if (V < 0 || B < 0 || 2 * V + B >= 0xFFFFFFFFLL) return bad("bad buffer sizes (need 2V+B < 2^32-1)");
    if (T < 0 || (T > 0 && !tri_table) || (((uintptr_t) tri_table) & 15)) return bad("bad triangle table (T rows of 16 B, 16-byte aligned)");
    memset(&F, 0, sizeof(F));
    F.tab = TriTable{tri_table, T};
    F.g.N = N;
    F.g.cam = cam;
    for (int k = 0; k < K; ++k) {
        const EpsmVertexRecord &v = verts[k];
        const EpsmScatterRecord &s = sc[k];
        if (!v.p0 || !v.p1 || !v.p2 || !v.n0 || !v.n1 || !v.n2 || !v.b0 || !v.b1 || !v.eta || !v.light ||
            !v.bsdf || !v.active || !v.active_em || !v.ismesh || !s.tri)
            return bad("NULL pointer in a vertex / scatter record");
        if (((((uintptr_t) s.aux) | ((uintptr_t) s.emit) | ((uintptr_t) s.shadow)) & 15) || (((uintptr_t) s.tri) & 3))
            return bad("aux/emit/shadow must be 16-byte aligned (tri: 4)");
        VertexPtrs<float> &o = F.g.v[k];
        o.p0 = (const float *) v.p0; o.p1 = (const float *) v.p1; o.p2 = (const float *) v.p2;
        o.n0 = (const float *) v.n0; o.n1 = (const float *) v.n1; o.n2 = (const float *) v.n2;
        o.b0 = (const float *) v.b0; o.b1 = (const float *) v.b1; o.eta = (const float *) v.eta;
        o.light = (const float *) v.light;
        o.bsdf = v.bsdf; o.active = v.active; o.active_em = v.active_em; o.ismesh = v.ismesh;
        ScatterPtrs<float> &t = F.s[k];
        t.tri = s.tri; t.aux = s.aux; t.emit = s.emit;
        t.shadow = k == 0 ? s.shadow : nullptr;       // epsm.py:610: `iteration == 0`
    }
    F.g.clip = (clip > 0.0f && clip <= 3.402823466e+38f) ? clip : 3.402823466e+38f;
    F.gpos = grad_pos; F.gnrm = grad_nrm; F.galpha = grad_alpha;
    F.V = V; F.B = grad_alpha ? B : 0;
    F.P = epsm_num_param_grads(variant, K);
    F.K = K;
    return EPSM_OK;
}

// ---- launch options: three process-wide integers, set from the environment by a static initialiser (never by an entry point)
namespace {
std::atomic<int64_t> g_options[3];
struct OptionsInit {
    OptionsInit() {
        const char *e = getenv("EPSM_SMALL_WAVEFRONT");
        g_options[EPSM_OPT_SMALL_WAVEFRONT_PATHS] = e ? atoll(e) : (int64_t) 1 << 20;
        const char *off = getenv("EPSM_NO_REPLICAS");
        g_options[EPSM_OPT_REPLICAS] = (off && off[0] == '1') ? 0 : 1;
        const char *one = getenv("EPSM_ONE_LAUNCH");
        g_options[EPSM_OPT_ONE_LAUNCH] = (one && one[0] == '1') ? 1 : 0;
    }
} g_options_init;
}  // namespace
namespace epsm {
int64_t fused_option(int option) { return g_options[option].load(std::memory_order_relaxed); }
}
extern "C" int epsm_set_option(int option, int64_t value) {
    epsm_host::err_buf()[0] = 0;
    if (option < 0 || option > EPSM_OPT_ONE_LAUNCH || value < 0) return fail(EPSM_EINVAL, "epsm_set_option: unknown option or negative value");
    g_options[option].store(value, std::memory_order_relaxed);
    return EPSM_OK;
}
extern "C" int64_t epsm_get_option(int option) {
    return (option < 0 || option > EPSM_OPT_ONE_LAUNCH) ? -1 : g_options[option].load(std::memory_order_relaxed);
}

extern "C" int epsm_release_workspace(void) {
    epsm_host::err_buf()[0] = 0;
    const hipError_t e = fused_release_workspaces();
    if (e != hipSuccess) return epsm_host::hip_fail("epsm_release_workspace", e);
    return EPSM_OK;
}

extern "C" int epsm_manifold_grad_scatter(int variant, int64_t N, int K,
                                          const float *cam, const EpsmVertexRecord *verts,
                                          const EpsmScatterRecord *sc, const uint32_t *tri_table, int64_t T,
                                          const float *dlduv, int64_t dlduv_stride, int dlduv_cols,
                                          const float *dldp, float clip,
                                          float *grad_pos, float *grad_nrm, float *grad_alpha,
                                          int64_t V, int64_t B, void *stream) {
    epsm_host::err_buf()[0] = 0;
    if (N == 0 && K >= 1 && K <= EPSM_MAX_VERTICES &&
        (variant == EPSM_VARIANT_MANIFOLD || variant == EPSM_VARIANT_MANIFOLD_CAUSTIC)) return EPSM_OK;
    FusedArgs F;
    const int rc = fill_args(F, "epsm_manifold_grad_scatter", variant, N, K, cam, verts, sc, tri_table, T, clip, grad_pos, grad_nrm, grad_alpha, V, B);
    if (rc != EPSM_OK) return rc;
    if (!dlduv || !dldp) return fail(EPSM_EINVAL, "epsm_manifold_grad_scatter: NULL argument");
    if (dlduv_cols < 0 || dlduv_stride < (dlduv_cols < 2 * K ? dlduv_cols : 2 * K))
        return fail(EPSM_EINVAL, "epsm_manifold_grad_scatter: dlduv_stride smaller than the columns to read");
    F.g.dlduv = dlduv;
    F.g.dlduv_stride = dlduv_stride;
    F.g.dldp = dldp;
    int dcols = dlduv_cols > 2 * K ? 2 * K : dlduv_cols;
    const bool full_d = dcols > 2;
    const hipError_t e = launch_backward_cp(variant, full_d ? kTangentsFullRows : kTangentsTwoColumns, false, F, dcols, (hipStream_t) stream);
    if (e != hipSuccess) return epsm_host::hip_fail("epsm_manifold_grad_scatter", e);
    return EPSM_OK;
}

extern "C" int epsm_backward_pass(int variant, int64_t N, int K, int64_t path_offset, int spp, int res,
                                  const float *ray_o, const float *ray_d, const float *ray_dx, const float *ray_dy,
                                  const float *grad_img, int img_width, int img_channels,
                                  const EpsmVertexRecord *verts, const EpsmScatterRecord *sc,
                                  const uint32_t *tri_table, int64_t T, float clip,
                                  float *grad_pos, float *grad_nrm, float *grad_alpha, float *grad_o_sum,
                                  int64_t V, int64_t B, void *stream) {
    epsm_host::err_buf()[0] = 0;
    if (N == 0 && K >= 1 && K <= EPSM_MAX_VERTICES &&
        (variant == EPSM_VARIANT_MANIFOLD || variant == EPSM_VARIANT_MANIFOLD_CAUSTIC)) return EPSM_OK;
    FusedArgs F;
    const int rc = fill_args(F, "epsm_backward_pass", variant, N, K, ray_o, verts, sc, tri_table, T, clip, grad_pos, grad_nrm, grad_alpha, V, B);
    if (rc != EPSM_OK) return rc;
    if (!ray_d || !ray_dx || !ray_dy || !grad_img) return fail(EPSM_EINVAL, "epsm_backward_pass: NULL argument");
    if (spp < 1 || res < 1 || img_width < res || img_channels < 5)
        return fail(EPSM_EINVAL, "epsm_backward_pass: need spp>=1, res>=1, img_width>=res, img_channels>=5");
    if (path_offset < 0 || (int64_t) res * res * spp < path_offset + N)
        return fail(EPSM_EINVAL, "epsm_backward_pass: path_offset + N exceeds res*res*spp");
    F.tin = TangentIn{path_offset, spp, res, img_width, img_channels, ray_o, ray_d, ray_dx, ray_dy, grad_img};
    F.grad_o_sum = grad_o_sum;
    const hipError_t e = launch_backward_cp(variant, kTangentsInKernel, false, F, 2, (hipStream_t) stream);
    if (e != hipSuccess) return epsm_host::hip_fail("epsm_backward_pass", e);
    return EPSM_OK;
}


// The same pass on the native log (include/epsm.h: EpsmPackedLog): one 128-byte record per (path, vertex), the
// rays and the flag word per path.
extern "C" int epsm_backward_pass_packed(int variant, int64_t N, int K, int64_t path_offset, int spp, int res,
                                         const EpsmPackedLog *log, const float *grad_img, int img_width, int img_channels,
                                         const uint32_t *tri_table, int64_t T, float clip,
                                         float *grad_pos, float *grad_nrm, float *grad_alpha, float *grad_o_sum,
                                         int64_t V, int64_t B, void *stream) {
    epsm_host::err_buf()[0] = 0;
    if (variant != EPSM_VARIANT_MANIFOLD && variant != EPSM_VARIANT_MANIFOLD_CAUSTIC) return fail(EPSM_EINVAL, "epsm_backward_pass_packed: unknown variant");
    if (K < 1 || K > EPSM_MAX_VERTICES) return fail(EPSM_EINVAL, "epsm_backward_pass_packed: K must be in 1..5");
    if (N == 0) return EPSM_OK;
    if (N < 0 || !log || !log->rays || !log->flags || !log->verts || !grad_img || !grad_pos || !grad_nrm)
        return fail(EPSM_EINVAL, "epsm_backward_pass_packed: NULL argument / bad N");
    if ((((uintptr_t) log->rays) | ((uintptr_t) log->verts) | ((uintptr_t) log->shadow) | ((uintptr_t) tri_table)) & 15)
        return fail(EPSM_EINVAL, "epsm_backward_pass_packed: rays / verts / shadow / tri_table must be 16-byte aligned");
    // (strides: 16-byte quads stay aligned; a window's 32-bit word offsets -- 2048 paths x stride -- stay far below 2^31)
    if (log->ray_stride < 0 || log->path_stride < 0 || (log->ray_stride & 3) || (log->path_stride & 3) || log->ray_stride > 4096 ||
        log->path_stride > 4096 || (log->ray_stride && log->ray_stride < 12) || (log->path_stride && log->path_stride < (int64_t) K * 32))
        return fail(EPSM_EINVAL, "epsm_backward_pass_packed: ray_stride / path_stride must be 0 or multiples of 4 words in [12 | 32 K, 4096]");
    if (V < 0 || B < 0 || 2 * V + B >= 0xFFFFFFFFLL || T < 0 || (T > 0 && !tri_table))
        return fail(EPSM_EINVAL, "epsm_backward_pass_packed: bad buffer sizes");
    if (spp < 1 || res < 1 || img_width < res || img_channels < 5)
        return fail(EPSM_EINVAL, "epsm_backward_pass_packed: need spp>=1, res>=1, img_width>=res, img_channels>=5");
    if (path_offset < 0 || (int64_t) res * res * spp < path_offset + N)
        return fail(EPSM_EINVAL, "epsm_backward_pass_packed: path_offset + N exceeds res*res*spp");
    FusedArgs F;
    memset(&F, 0, sizeof(F));
    F.tab = TriTable{tri_table, T};
    F.g.N = N;
    F.g.clip = (clip > 0.0f && clip <= 3.402823466e+38f) ? clip : 3.402823466e+38f;
    F.gpos = grad_pos; F.gnrm = grad_nrm; F.galpha = grad_alpha;
    F.V = V; F.B = grad_alpha ? B : 0;
    F.P = epsm_num_param_grads(variant, K);
    F.K = K;
    F.pk_rays = log->rays; F.pk_flags = log->flags; F.pk_verts = (const float *) log->verts; F.pk_shadow = log->shadow;
    F.pk_ray_stride = log->ray_stride ? log->ray_stride : 12;
    F.pk_path_stride = log->path_stride ? log->path_stride : (int64_t) K * kRecWords;
    if ((log->path_list != nullptr) != (log->path_count != nullptr) || (log->path_list && grad_o_sum))
        return fail(EPSM_EINVAL, "epsm_backward_pass_packed: path_list and path_count come together, and with grad_o_sum = NULL");
    F.pk_list = log->path_list; F.pk_list_count = log->path_count;
    F.tin = TangentIn{path_offset, spp, res, img_width, img_channels, nullptr, nullptr, nullptr, nullptr, grad_img};
    F.grad_o_sum = grad_o_sum;
    const hipError_t e = launch_backward_cp(variant, kTangentsInKernel, true, F, 2, (hipStream_t) stream);
    if (e != hipSuccess) return epsm_host::hip_fail("epsm_backward_pass_packed", e);
    return EPSM_OK;
}
