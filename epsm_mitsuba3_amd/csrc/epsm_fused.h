// epsm_fused.h -- what the two fused backward kernels share: launch arguments, the native record layout, the
// replica workspace of small wavefronts.  (epsm_grad_scatter.hip: one lane per path, epsm_path_core.h;
// epsm_backward_cp.hip: one lane per (path, constraint vertex), epsm_cp_core.h.)
#pragma once

#include "epsm_common.h"
#include "epsm_path_core.h"
#include "epsm_scatter_core.h"
#include "epsm_tangent_core.h"

namespace epsm {

struct FusedArgs {
    GradArgs<float> g;
    ScatterPtrs<float> s[kMaxVertices];
    TriTable tab;                    // the scene's triangles: id -> [v0, v1, v2, mode]
    float *gpos, *gnrm, *galpha;
    int64_t V, B;
    int P, K;
    TangentIn tin;                   // epsm_backward_pass: the first-vertex tangent is computed in the kernel
    float *grad_o_sum;
    // small wavefronts: workgroup b adds to replica b % replicas of the four buffers (launch(), reduce_replicas_kernel)
    float *rep;                      // replicas x rep_stride floats, each [pos 3V | nrm 3V | alpha B | o_sum 3]; null: none
    int replicas;
    int64_t rep_stride;
    uint32_t *rep_done;              // per replica: workgroups that have flushed into it; non-null = the LAST of them adds the replica to the
                                     // caller's buffers and clears it (one launch); null = reduce_replicas_kernel does (two launches)
    int rep_blocks;                  // workgroups of the launch (b % replicas == r of them feed replica r)
    // epsm_backward_pass_packed: the native log (include/epsm.h, EpsmPackedLog) instead of the per-array records
    const float *pk_rays;            // (N,12)  o, d, d_x, d_y
    const uint32_t *pk_flags;        // (N)     5 bits per vertex
    const float *pk_verts;           // (N,K,32) one 128-byte record per (path, vertex)
    const uint32_t *pk_shadow;       // (N,4) or null
    int64_t pk_ray_stride, pk_path_stride;   // words between consecutive paths' rays / first records (dense: 12, 32 K)
    const uint32_t *pk_list, *pk_list_count; // EpsmPackedLog.path_list / path_count: the windows run over these paths (null: over all N)
};
// where the tangents of a path come from
enum { kTangentsTwoColumns = 0, kTangentsFullRows = 1, kTangentsInKernel = 2 };

// The 85 array pointers of a K = 5 launch are 170 SGPRs: kept as kernel arguments the compiler hoists them
// out of the persistent loop and spills ~240 of them into VGPR lanes (a v_readlane per use: ~10 % of the
// kernel's VALU instructions).  The workgroup copies them to LDS once; the path code reads the few it
// needs per vertex with ds_read_b64 into VGPR pairs that die with the loads they feed.
struct PtrTable {
    VertexPtrs<float> v[kMaxVertices];
    ScatterPtrs<float> s[kMaxVertices];
};
// 16-byte global load of quad q of a packed vertex record
typedef float F4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ F4v ldq(const float *rec, int q) {
    return *(const __attribute__((address_space(1))) F4v *) (rec + 4 * q);
}
// EpsmPackedLog vertex record (32 words, include/epsm.h): p0 p1 p2 | b0 b1 tri | n0 eta || n1 n2 | light.xy | etri eb0 eb1 ew |
// light.z dhf(3) -- the first 64-byte sector holds all an END point or a diffuse first hit is read for
constexpr int kRecWords = 32;
__device__ __forceinline__ Geo<float> geo_from(F4v q0, F4v q1, F4v q2) {
    Geo<float> g;
    const V3<float> p0 = mk3<float>(q0.x, q0.y, q0.z), p1 = mk3<float>(q0.w, q1.x, q1.y), p2 = mk3<float>(q1.z, q1.w, q2.x);
    g.b0 = q2.y; g.b1 = q2.z;
    g.x = p0 * g.b0 + p1 * g.b1 + p2 * (1.f - g.b0 - g.b1);
    g.e1 = p0 - p2; g.e2 = p1 - p2;
    return g;
}
__device__ __forceinline__ Nrm<float> nrm_from(F4v q3, F4v q4, F4v q5, float b0, float b1) {
    Nrm<float> o;
    const V3<float> n0 = mk3<float>(q3.x, q3.y, q3.z), n1 = mk3<float>(q4.x, q4.y, q4.z), n2 = mk3<float>(q4.w, q5.x, q5.y);
    o.n = n0 * b0 + n1 * b1 + n2 * (1.f - b0 - b1);
    o.dn1 = n0 - n2; o.dn2 = n1 - n2;
    return o;
}

// epsm.py:250-272 on the packed log: rays = this path's 12 floats (o, d, d_x, d_y), rec1 = its first vertex record
__device__ __forceinline__ Tangent first_vertex_tangent_packed(const TangentIn &A, int64_t i, const float *rays,
                                                               const float *rec1, bool active) {
    const int64_t pix = (A.path_offset + i) / A.spp;
    const int64_t y = pix / A.res, x = pix % A.res;
    const auto *g = gl(A.grad_img) + (y * A.img_width + x) * A.img_channels;
    const float gx = g[3], gy = g[4];
    const F4v r0 = ldq(rays, 0), r1 = ldq(rays, 1), r2 = ldq(rays, 2);
    const V3<float> o = mk3<float>(r0.x, r0.y, r0.z), d = mk3<float>(r0.w, r1.x, r1.y), dx = mk3<float>(r1.z, r1.w, r2.x),
                    dy = mk3<float>(r2.y, r2.z, r2.w);
    V3<float> p0 = zero3<float>(), p1 = p0, p2 = p0;
    if (active) {
        const F4v q0 = ldq(rec1, 0), q1 = ldq(rec1, 1), q2 = ldq(rec1, 2);
        p0 = mk3<float>(q0.x, q0.y, q0.z); p1 = mk3<float>(q0.w, q1.x, q1.y); p2 = mk3<float>(q1.z, q1.w, q2.x);
    }
    return tangent_from(o, d, dx, dy, gx, gy, p0, p1, p2, active);
}


// ---- small wavefronts: replicas of the gradient buffers (epsm_grad_scatter.hip)
constexpr size_t kReplicaBudget = 48u << 20;
hipError_t fused_workspace(hipStream_t s, size_t bytes, float **out);       // kReplicaBudget bytes of replicas + 4 KB of counters behind them
hipError_t fused_release_workspaces();
void fused_workspace_invalidate(hipStream_t s);     // after a failed launch: this stream's workspace is re-allocated and zeroed before its next use
__global__ void reduce_replicas_kernel(float *rep, int replicas, int64_t stride, int64_t V, int64_t B,
                                       float *gpos, float *gnrm, float *galpha, float *go);

// ---- launch options (include/epsm.h, epsm_set_option): process-wide, initialised from the environment when the library loads
int64_t fused_option(int option);

// ---- the constraint-parallel form (epsm_backward_cp.hip).  dmode: kTangents*; packed: the native log.
hipError_t launch_backward_cp(int variant, int dmode, bool packed, const FusedArgs &F, int dcols, hipStream_t s);

}  // namespace epsm
