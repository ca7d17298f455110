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

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EPSM_ABI_VERSION 1

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

/* Human-readable text of the last failure on the calling thread ("" if none). */
const char *epsm_last_error(void);

/* ABI version of the loaded library (EPSM_ABI_VERSION at build time). */
int epsm_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* EPSM_H */
