/*
 * epsm.h -- C ABI of the MI355X-native EPSM manifold-gradient hot path.
 *
 * Every entry point replaces one piece of the reference's Python/PyTorch hot
 * path in  src/python/python/ad/integrators/epsm.py  (jkxing/EPSM_Mitsuba3);
 * the reference has no FFI of its own for this path (it is Python + torch +
 * Dr.Jit), so the boundary is drawn where a maintainer would bind a native
 * library: at the tensors that cross between Dr.Jit/torch and the per-path
 * arithmetic.  INTEGRATION.md shows the ctypes stub for each call.
 *
 * Conventions (all entry points)
 *   - plain pointers + sizes, no torch / Dr.Jit types;
 *   - "device" pointers are HIP device pointers (hipMalloc / torch-ROCm
 *     storage); the oracle twin in oracle/ takes host pointers with the same
 *     struct layout;
 *   - vectors are stored exactly as the reference's torch tensors are:
 *     (N,3) row-major fp32 ("AoS of 3"), scalars (N) fp32, masks (N) u8;
 *   - no allocation, no host synchronisation, no exceptions: the launch is
 *     enqueued on `stream` (a hipStream_t passed as void*, NULL = default
 *     stream) and the call returns 0, or a negative EPSM_E* code;
 *   - thread-safe for distinct streams.
 */
#ifndef EPSM_H
#define EPSM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Changes when the meaning or signature of an existing entry point changes (4: triangle-id addressing, round 2; 5: word layout of
 * the native log's vertex record, round 4 -- EpsmPackedLog below).  Entry points added since without touching the others:
 * epsm_backward_pass_packed, epsm_release_workspace, epsm_sinkhorn_*, epsm_set_option / epsm_get_option, epsm_probe.
 * 6 (round 5): EpsmEnvironment starts with `kind` (0 = none: a zeroed EpsmScene has no environment; until 5 `emitter = -1` said
 * so); note of 5, late: epsm_trace_paths_reparam had gained its `flags` parameter mid-signature in that version.
 * 7 (round 5): EpsmPackedLog ends with ray_stride / path_stride, EpsmRecordOut (epsm_trace.h) with ray_stride / packed_stride: the
 * native log may interleave a path's rays and records in one block (zeroed strides = the dense arrays of version 6); EpsmPackedLog
 * also ends with path_list / path_count (NULL = all N paths), EpsmRecordOut with first_hit (EPSM_TRACE_FUSE_FIRST_HIT, epsm_trace.h). */
#define EPSM_ABI_VERSION 7

/* BSDF flag bits tested by the hot path (include/mitsuba/render/bsdf.h:40-46,101). */
#define EPSM_BSDF_NULL     0x1u
#define EPSM_BSDF_DIFFUSE  0x6u

/* Largest number of logged surface vertices per path.  The reference logs at
 * most 5 (epsm.py:648 `iteration<5`). */
#define EPSM_MAX_VERTICES 5

enum {
    EPSM_VARIANT_MANIFOLD = 0,         /* ManifoldIntegrator.calc_grad        epsm.py:745-946  */
    EPSM_VARIANT_MANIFOLD_CAUSTIC = 1  /* ManifoldCausticIntegrator.calc_grad epsm.py:952-1200 */
};

enum {
    EPSM_OK = 0,
    EPSM_EINVAL = -22,     /* bad argument (NULL pointer, K out of range, ...)        */
    EPSM_ENODEV = -19,     /* no HIP device / kernel image not loadable on this device */
    EPSM_ELAUNCH = -5      /* hipLaunch / hip runtime error (see epsm_last_error())    */
};

/* One logged path vertex, as recorded by EPSMIntegrator.sample_path
 * (epsm.py:648-654) from the patched SurfaceInteraction / BSDFSample fields
 * (include/mitsuba/render/interaction.h:221-224, bsdf.h:197).
 * All pointers address N elements (paths) of the stated shape. */
typedef struct EpsmVertexRecord {
    const void *p0, *p1, *p2;   /* (N,3) f32  triangle vertex positions  "points"[0..2] */
    const void *n0, *n1, *n2;   /* (N,3) f32  triangle vertex normals    "normals"[0..2] */
    const void *b0, *b1;        /* (N)   f32  barycentric weights of p0,p1  "uv"[0..1]   */
    const void *eta;            /* (N)   f32  bsdf_sample.eta                           */
    const void *hf;             /* (N,3) f32  sampled microfacet normal (may be NULL: the
                                              gradients do not depend on its value)     */
    const void *light;          /* (N,3) f32  emitter sample position ds.p              */
    const uint32_t *bsdf;       /* (N)   u32  bsdf.flags()                              */
    const uint8_t *active;      /* (N)   u8   active & si.is_valid()                    */
    const uint8_t *active_em;   /* (N)   u8   emitter sample usable                     */
    const uint8_t *ismesh;      /* (N)   u8   si.ismesh > 0                             */
} EpsmVertexRecord;

/* Number of (N,3) parameter-gradient arrays calc_grad returns:
 * 5K for "manifold" (epsm.py:786-788,815-816), 5K-2 for "manifold_caustic"
 * (n,m are registered only when a continuing sub-path exists, epsm.py:1102-1105). */
int epsm_num_param_grads(int variant, int K);

/* ---------------------------------------------------------------------------
 * epsm_manifold_grad  --  replaces  {Manifold,ManifoldCaustic}Integrator.calc_grad
 *                         (epsm.py:745 / 952; called from render_backward, epsm.py:275)
 *
 *   variant      EPSM_VARIANT_*
 *   N            number of paths (wavefront size)
 *   K            logged surface vertices per path, 1..EPSM_MAX_VERTICES
 *                (= len(path_info) - 1)
 *   cam          (N,3) f32   path_info[0]["cam"]
 *   verts        K records (host array of structs holding device pointers)
 *   dlduv        f32, row n starts at dlduv + n*dlduv_stride and holds the
 *                tangent of the barycentrics, column 2(k-1)+j = b_j of vertex k
 *                (the reference passes (N,1,2L): stride 2L).  Only the first
 *                `dlduv_cols` columns are read; the rest are taken as zero
 *                (render_backward only ever fills columns 0,1: epsm.py:256,268-269).
 *   dldp         (N,3) f32   tangent of the first hit point (epsm.py:270)
 *   clip         outlier threshold of epsm.py:932-944 (0.1 in the reference);
 *                <= 0 or +inf disables the clamp
 *   out_param    (P,N,3) f32, P = epsm_num_param_grads(); entry 5(k-1)+{0,1,2,3,4}
 *                = gradient w.r.t. p0,p1,p2,n,m of vertex k   ("final_param_grad")
 *   out_light    (K,N,3) f32  "light_grad"
 *   out_diffuse  (K,N,3) f32  "diffuse_grad"  (entry 0 = masked dldp, epsm.py:791-792)
 *
 * Inputs are not modified (the reference zeroes rows of dldp / dlduv in place,
 * epsm.py:791,999; the results are identical).
 * ------------------------------------------------------------------------- */
int epsm_manifold_grad(int variant, int64_t N, int K,
                       const float *cam, const EpsmVertexRecord *verts,
                       const float *dlduv, int64_t dlduv_stride, int dlduv_cols,
                       const float *dldp, float clip,
                       float *out_param, float *out_light, float *out_diffuse,
                       void *stream);

/* ---------------------------------------------------------------------------
 * epsm_first_vertex_tangent  --  replaces the forward-mode AD block of
 *     EPSMIntegrator.render_backward, epsm.py:238-272:
 *         grad_d = (d_x - d) * gx + (d_y - d) * gy            (epsm.py:250-255)
 *         set_grad(ray.d, grad_d); re-intersect; forward_to(si.p)
 *         dlduv[:,0,0] = grad(si.b0); dlduv[:,0,1] = grad(si.b1); dldp1 = grad(si.p)
 *     with the Moeller-Trumbore intersection of include/mitsuba/render/mesh.h:343-365
 *     and b1 = u, b2 = v, b0 = 1-u-v, p = p0 b0 + p1 b1 + p2 b2 (src/render/mesh.cpp:698-709).
 *
 *   N, spp       paths are ordered (pixel, sample): pixel = (path_offset + n) / spp,
 *                row-major (the reference reshapes to (res,res,spp,3), epsm.py:250);
 *   path_offset  index of this call's first path within the whole wavefront (0 when
 *                the wavefront is processed in one piece; tiles / GPU shards pass
 *                their start)
 *   res          side of the backward sensor's film; pixel -> (y,x) = (pix / res, pix % res)
 *   ray_o/d/dx/dy (N,3) f32  primary ray origin, direction and the two one-pixel
 *                offset directions of sample_ray_differential
 *   grad_img     gradient image, row-major (rows, img_width, img_channels) f32; the
 *                top-left res x res crop is used (epsm.py:240) and channels 3,4 are
 *                the image-space motion (gx, gy) (epsm.py:254)
 *   p0,p1,p2     (N,3) f32 triangle of the first hit; active (N) u8 (0 -> outputs 0)
 *   dlduv        written: row n at dlduv + n*dlduv_stride gets (d b0, d b1) in
 *                columns 0,1; columns 2..dlduv_stride-1 are zero-filled
 *   dldp         (N,3) written: d si.p
 *   grad_o_sum   optional (3 floats, accumulated atomically): sum_n -grad_d, the
 *                camera-origin gradient of epsm.py:260-261; may be NULL
 * ------------------------------------------------------------------------- */
int epsm_first_vertex_tangent(int64_t N, int64_t path_offset, int spp, int res,
                              const float *ray_o, const float *ray_d,
                              const float *ray_dx, const float *ray_dy,
                              const float *grad_img, int img_width, int img_channels,
                              const float *p0, const float *p1, const float *p2,
                              const uint8_t *active,
                              float *dlduv, int64_t dlduv_stride, float *dldp,
                              float *grad_o_sum, void *stream);

/* Per-vertex addressing of the scene-parameter buffers, logged by the tracer next
 * to EpsmVertexRecord.  Replaces the AD graph that Dr.Jit records through
 * Mesh::vertex_position / vertex_normal gathers (include/mitsuba/render/mesh.h:94-106). */
#define EPSM_NO_INDEX 0xFFFFFFFFu
#define EPSM_MODE_VERTEX_NORMALS 0x1u  /* mesh has vertex normals (mesh.cpp:784-790), else flat (mesh.cpp:811-816) */
#define EPSM_MODE_FLIP_NORMALS   0x2u  /* m_flip_normals (mesh.cpp:820-827) */
#define EPSM_MODE_POS_ATTACHED   0x4u  /* vertex positions of this mesh receive gradients */
#define EPSM_MODE_NRM_ATTACHED   0x8u  /* vertex normals of this mesh receive gradients */

/* The scene's triangles: row t = [v0, v1, v2, mode] -- the rows of triangle t's vertices in the flat (V,3) position /
 * normal buffers and, in the mode word, the EPSM_MODE_* bits of its mesh (bits 0..3) and the alpha slot of the mesh's
 * BSDF + 1 (bits 8.., 0 = no attached BSDF parameter; read by epsm_backward_pass_packed, whose log has no per-vertex
 * `aux` slot).  One table per scene (16 B per triangle; it stays in L2 / MALL),
 * rebuilt when a mesh is attached or detached.  A ray tracer reports a hit as a primitive index, so the per-path log
 * below carries triangle IDS (4 B) instead of vertex triples.
 * Passed to the scatter entry points as `tri_table` ((T,4) u32, 16-byte aligned) and `T`. */

typedef struct EpsmScatterRecord {
    const uint32_t *tri;    /* (N)   u32  id of the hit triangle = row of the triangle table (EPSM_NO_INDEX or any
                                          id >= T = none: analytic shape, detached lane) */
    const uint32_t *aux;    /* (N,4)      [bsdf_id u32, d hf / d alpha as 3 x f32 bits]: slot in grad_alpha
                                          (EPSM_NO_INDEX = none) and the derivative of the sampled microfacet normal
                                          w.r.t. the BSDF's alpha (roughconductor.cpp:249-255); may be NULL (no BSDF
                                          parameter attached) */
    const uint32_t *emit;   /* (N,4)      [etri u32, eb0, eb1, eweight f32]: triangle hit by the emitter-sample shadow
                                          ray (epsm.py:622-625), its barycentrics and sum_rgb(Lr_dir) (epsm.py:627);
                                          may be NULL */
    const uint32_t *shadow; /* (N,4)      [stri u32, sb0, sb1, dis f32]: read for the FIRST logged vertex only (sc[0]).
                                          The occluder term of epsm.py:609-620 (integrators with max_depth <= 3): first
                                          surface hit by the ray from the first vertex towards its emitter sample
                                          (closest hit, no maximum distance), its barycentrics and
                                          dis = |ds.p - hit| / |ds.p - si.p| (0 when < 0.01, :614-615); that triangle
                                          receives diffuse_grad[0] * dis * barycentric (:616-618).  May be NULL */
} EpsmScatterRecord;

/* ---------------------------------------------------------------------------
 * epsm_scatter  --  replaces the Backward-mode replay of sample_path
 *     (epsm.py:283-297 -> 559-562, 622-627, 644-645): the per-path gradients of
 *     calc_grad are accumulated into the parameter-gradient buffers, i.e. the
 *     adjoint of the vertex gathers (mesh.h:94-106) as float atomics.
 *   per logged vertex k (iteration it = k-1):
 *     (vidx = table[tri][0..2], mode = table[tri][3], evidx = table[emit[0]][0..2], eb = emit[1..2],
 *      eweight = emit[3], bsdf_id = aux[0], dhf_dalpha = aux[1..3])
 *     grad_pos[vidx_j] += out_param[5it+j]                      (si.p_j * path_grad[5it+j],  :559-560)
 *     grad_pos[vidx_j] += b_j * out_diffuse[it]                 (si_follow.p * diffuse_grad[it], :561-562)
 *     normals:  d/dn_j of  sh_frame.n . out_param[5it+3]        (:645; mesh.cpp:784-790 or flat :729,811-816)
 *     grad_alpha[bsdf_id] += dhf_dalpha . out_param[5it+4]      (bsdf_sample.hf * path_grad[5it+4], :645)
 *     grad_pos[evidx_j] += eb_j * out_light[it] * eweight       (si_direct.p * light_grad[it] * sum Lr_dir, :626-627;
 *                                                                only when the emitter triangle's mesh has
 *                                                                EPSM_MODE_POS_ATTACHED, as for every position row)
 *     it = 0 only, with sc[0].shadow = [stri, sb0, sb1, dis] and s_j = table[stri][j]:
 *     grad_pos[s_j]     += sb_j * dis * out_diffuse[0]          (si_direct.p * diffuse_grad[0] * dis, :616-618)
 *   (for "manifold_caustic" the last vertex has no n,m entries.)
 *   grad_pos / grad_nrm: (V,3) f32, grad_alpha: (B) f32; accumulated, not cleared.
 * ------------------------------------------------------------------------- */
int epsm_scatter(int variant, int64_t N, int K,
                 const EpsmVertexRecord *verts, const EpsmScatterRecord *sc, const uint32_t *tri_table, int64_t T,
                 const float *out_param, const float *out_light, const float *out_diffuse,
                 float *grad_pos, float *grad_nrm, float *grad_alpha,
                 int64_t V, int64_t B, void *stream);

/* ---------------------------------------------------------------------------
 * epsm_manifold_grad_scatter  --  epsm_manifold_grad + epsm_scatter in ONE launch:
 *     calc_grad (epsm.py:275) followed by the Backward-mode replay (epsm.py:283-297)
 *     without materialising final_param_grad / light_grad / diffuse_grad in HBM.
 *     Arguments as in the two calls above; results are ACCUMULATED into
 *     grad_pos / grad_nrm (V,3) and grad_alpha (B) with float atomics (the sums are
 *     those of the two-call form up to the order of the additions; the kernel regroups
 *     paths inside 1024-path windows, so the order also differs from run to run).  This is what
 *     EPSMIntegrator.render_backward uses; the two-call form exists for callers that
 *     want calc_grad's lists (the drop-in of INTEGRATION.md section 1).
 *     Range of the sums (this and the two entry points below): with clip <= 1 the rows of a window of <= 2048 paths are summed
 *     on chip in 64-bit fixed point (44 fractional bits, |sum| < 2^19); only terms of magnitude < 16 enter those rows -- a path
 *     adds at most 16 terms to one row, 2048 x 16 x 16 = 2^19 is never reached, the integer cannot wrap -- larger ones (a
 *     clamped term times an emitter weight or 1 / |n| beyond ~10^2) are added to the caller's buffers directly as float atomics,
 *     non-finite ones add nothing.  With clip > 1 (the clamp switched off) the rows are float.
 * ------------------------------------------------------------------------- */
int epsm_manifold_grad_scatter(int variant, int64_t N, int K,
                               const float *cam, const EpsmVertexRecord *verts,
                               const EpsmScatterRecord *sc, const uint32_t *tri_table, int64_t T,
                               const float *dlduv, int64_t dlduv_stride, int dlduv_cols,
                               const float *dldp, float clip,
                               float *grad_pos, float *grad_nrm, float *grad_alpha,
                               int64_t V, int64_t B, void *stream);

/* ---------------------------------------------------------------------------
 * epsm_backward_pass  --  epsm_first_vertex_tangent + epsm_manifold_grad_scatter in ONE launch:
 *     everything render_backward does between the trace and the parameter gradients
 *     (epsm.py:238-297).  The tangents dlduv / dldp1 are computed per path in registers
 *     instead of being written and read back; cam of epsm_manifold_grad is ray_o
 *     (epsm.py:547); grad_o_sum (3 floats, may be NULL) receives -sum grad_d (epsm.py:260-261).
 *     Arguments as in the two calls it replaces.
 * ------------------------------------------------------------------------- */
int epsm_backward_pass(int variant, int64_t N, int K, int64_t path_offset, int spp, int res,
                       const float *ray_o, const float *ray_d, const float *ray_dx, const float *ray_dy,
                       const float *grad_img, int img_width, int img_channels,
                       const EpsmVertexRecord *verts, const EpsmScatterRecord *sc, const uint32_t *tri_table, int64_t T, float clip,
                       float *grad_pos, float *grad_nrm, float *grad_alpha, float *grad_o_sum,
                       int64_t V, int64_t B, void *stream);

/* ---------------------------------------------------------------------------
 * The NATIVE path log: what this library's tracer writes for this library's backward kernel (the per-array records
 * above are the reference's tensors, kept for the calc_grad drop-in).  The backward kernel regroups the paths of a
 * window by chain length, so a wave reads the records of 64 scattered paths: with one array per field that is nine
 * 12-byte gathers per vertex from lines of which a fraction is wanted; here a (path, vertex) is ONE 128-byte record,
 * a path's K records are contiguous, and vertices a path never reached are never touched (nor written).
 *   rays   (N,12) f32   o, d, d_x, d_y of the primary ray (sample_ray_differential)
 *   flags  (N)    u32   5 bits per logged vertex, vertex k at bits 5(k-1)..: 1 = bsdf has a Diffuse lobe, 2 = Null,
 *                       4 = active, 8 = active_em, 16 = ismesh  (the masks of EpsmVertexRecord)
 *   verts  (N,K,32) 32-bit words, record of vertex k of path i at word (i*K + k-1)*32 (16-byte aligned):
 *            0..8   p0 p1 p2       9 10  b0 b1       11  tri u32 (row of the triangle table)       12..14  n0      15  eta
 *            16..21 n1 n2          22 23 light.x .y  24..27 etri u32, eb0, eb1, eweight (EpsmScatterRecord.emit)
 *            28     light.z        29..31 d hf / d alpha
 *          -- words 0..15 are one 64-byte sector of the record's cache line: all that is read of a vertex that is only the END
 *          point of a chain or a diffuse first hit (geometry, barycentrics, triangle id) comes with ONE sector (ABI v5; v4 kept
 *          b0, b1 at 18, 19 and tri at 28: two sectors for such a vertex)
 *          the alpha slot of the vertex's BSDF travels with the triangle: bits 8.. of the table row's mode word hold
 *          slot + 1 (0 = none)
 *   shadow (N,4) as EpsmScatterRecord.shadow, or NULL
 *   ray_stride, path_stride (ABI v7)   words between the rays / the first records of consecutive paths; 0 = the dense arrays above
 *          (12 and 32 K).  An INTERLEAVED form for hosts that keep one block per path: 32 (K + 1) words = K + 1 cache lines,
 *          128-byte aligned --
 *              words 0..11 the rays, 12..15 free | 16..: the K records          (rays = block, verts = block + 16, both strides 32 (K + 1))
 *          -- which puts the records HALF a line off the lines: line 0 of a path = [rays | first sector of vertex 1], line k =
 *          [second sector of vertex k | first sector of vertex k + 1].  What a path's lanes read is then a run of WHOLE lines: a
 *          diffuse first hit = line 0 alone (rays + geometry; the dense form: a line of rays and half a record line), a chain of
 *          m constraint vertices and its end point = lines 0..m, every byte of which is wanted (the dense form: a line of rays, m
 *          record lines and half of one more for the end point's first sector).  Same words, same records; only where they lie.
 *          Measured on MI355X (MEASUREMENTS.md 10.12): 16 % fewer bytes from HBM, 1 % less kernel time on the headline slab -- the
 *          kernel is bound by its vector instructions, not by those bytes -- and a tracer whose 48-byte ray stores no longer
 *          coalesce (trace + log + 8 %): this library's tracer and benchmark keep the dense arrays.
 * epsm_backward_pass_packed = epsm_backward_pass on this log (same sums, same arguments otherwise).
 * ------------------------------------------------------------------------- */
typedef struct EpsmPackedLog {
    const float *rays;
    const uint32_t *flags;
    const void *verts;
    const uint32_t *shadow;
    int64_t ray_stride, path_stride;   /* words; 0 = 12 / 32 K */
    const uint32_t *path_list;         /* NULL, or (<= N) indices, ascending, of the only paths the backward pass is to look at -- every other
                                          path's flag word is 0 (EPSM_TRACE_FUSE_FIRST_HIT: the tracer's list of survivors); needs grad_o_sum = NULL */
    const uint32_t *path_count;        /* device word: how many entries path_list holds (read by the kernel: no host round trip) */
} EpsmPackedLog;
#define EPSM_FLAG_DIFFUSE   1u
#define EPSM_FLAG_NULL      2u
#define EPSM_FLAG_ACTIVE    4u
#define EPSM_FLAG_ACTIVE_EM 8u
#define EPSM_FLAG_ISMESH    16u

int epsm_backward_pass_packed(int variant, int64_t N, int K, int64_t path_offset, int spp, int res,
                              const EpsmPackedLog *log, const float *grad_img, int img_width, int img_channels,
                              const uint32_t *tri_table, int64_t T, float clip,
                              float *grad_pos, float *grad_nrm, float *grad_alpha, float *grad_o_sum,
                              int64_t V, int64_t B, void *stream);

/* Small wavefronts (N <= 2^20 paths) of the three fused entry points above -- epsm_manifold_grad_scatter,
 * epsm_backward_pass, epsm_backward_pass_packed -- accumulate into REPLICAS of the gradient buffers, which one more small
 * kernel on the same stream sums into grad_pos / grad_nrm / grad_alpha / grad_o_sum ("Small wavefronts": the
 * workgroups of a small launch all flush at once, and same-address global atomics retire one at a time).  The replicas
 * live in a workspace of 48 MB that the library allocates on first use per (device, stream) -- at most 16 of them; further
 * streams go without replicas -- and keeps zeroed between launches.  Consequences for the caller: the first small launch on
 * a stream calls hipMalloc (do it outside a stream capture); results are complete when the stream reaches the end of the
 * call's work, as before; launches on ONE stream from several host threads must be serialised by the caller, as any use
 * of a stream.  epsm_release_workspace frees every workspace (call it when no launch is in flight; they come back on
 * demand).
 * There is no reference counterpart: Dr.Jit's scatter_reduce goes straight to global atomics. */
int epsm_release_workspace(void);

/* Launch options of the three fused entry points (process-wide; a launch reads them when it is issued).  Their initial values
 * come from the environment ONCE, when the library is loaded -- no entry point calls getenv.
 *   EPSM_OPT_SMALL_WAVEFRONT_PATHS  wavefronts of up to this many paths take the small form (windows of 128 .. 1024 paths,
 *                                   replicas); larger ones walk windows of 2048 paths.  Default 2^20; EPSM_SMALL_WAVEFRONT=<paths>.
 *   EPSM_OPT_REPLICAS               1: small wavefronts accumulate into replicas (above); 0: straight into the caller's
 *                                   buffers.  Default 1; EPSM_NO_REPLICAS=1 sets 0.
 *   EPSM_OPT_ONE_LAUNCH             0: the replicas are summed by a second small kernel on the same stream; 1: inside the launch --
 *                                   the last workgroup to flush into a replica adds it to the caller's buffers and clears it.
 *                                   Default 0 (EPSM_ONE_LAUNCH=1 sets 1): the agent-scope release every workgroup then needs
 *                                   before it is counted is an L2 write-back on this eight-XCD part -- 0.144 ms against 0.086 ms
 *                                   with two launches on the reference's own backward size (524 288 paths, K = 2).
 * Results are the same sums either way (the parity tests run both forms at every size).
 * epsm_set_option returns EPSM_OK or -EINVAL (unknown option, negative value); epsm_get_option returns -1 for an unknown option. */
#define EPSM_OPT_SMALL_WAVEFRONT_PATHS 0
#define EPSM_OPT_REPLICAS 1
#define EPSM_OPT_ONE_LAUNCH 2
int epsm_set_option(int option, int64_t value);
int64_t epsm_get_option(int option);

/* The Sinkhorn matcher's inner operation (outer loop).  EPSM/utils/matcher.py:51-63 calls
 * geomloss.SamplesLoss("sinkhorn", p=2, blur=0.01, scaling=0.9) on two 5-D point clouds (r,g,b,x,y); geomloss's online
 * backend evaluates, per dual update, the "softmin"
 *     out[i]  = -eps * log sum_j exp( h[j] - |x_i - y_j|^2 / (2 eps) )                    i < n, j < m
 *     wsum[i] = sum_j p_ij y_j,  p_ij = softmax_j( h[j] - |x_i - y_j|^2 / (2 eps) )        (optional, (n,D); d out[i] / d x_i = x_i - wsum[i])
 * without forming the n x m cost matrix.  x (n,D), y (m,D) row-major floats, 1 <= D <= 7, h (m) = log-weight + dual / eps;
 * scratch: device memory of at least epsm_sinkhorn_scratch_bytes(n, m, D) bytes (partial results of the column ranges
 * a row is split into: epsm_sinkhorn_splits).  epsm_mitsuba3_amd/matcher.py drives it (epsilon-scaling loop, debiasing). */
int epsm_sinkhorn_splits(int64_t n, int64_t m);
size_t epsm_sinkhorn_scratch_bytes(int64_t n, int64_t m, int D);
int epsm_sinkhorn_softmin(int64_t n, int64_t m, int D, const float *x, const float *y, const float *h, float eps,
                          float *out, float *wsum, void *scratch, size_t scratch_bytes, void *stream);
/* One dual update of the Sinkhorn loop in one call: h[j] = log_weight + dual[j] / eps (dual NULL: h = log_weight, the
 * initialisation), out[i] = prev ? (prev[i] + softmin_i) / 2 : softmin_i (the symmetric, averaged update of geomloss's
 * sinkhorn_loop).  `out` is written by the second of the two kernels, after every read of `dual` and element-wise after
 * `prev`: it may alias either (the caller still needs the OLD dual of the other cloud for its own update). */
int epsm_sinkhorn_update(int64_t n, int64_t m, int D, const float *x, const float *y, const float *dual, float log_weight,
                         float eps, const float *prev, float *out, float *wsum, void *scratch, size_t scratch_bytes, void *stream);

/* Human-readable text of the last failure on the calling thread ("" if none). */
const char *epsm_last_error(void);

/* ABI version of the loaded library (EPSM_ABI_VERSION at build time). */
int epsm_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* EPSM_H */
