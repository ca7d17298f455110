/*
 * epsm_trace.h -- C ABI of the wavefront path tracer that PRODUCES the EPSM path records.
 *
 * Replaces, for triangle-mesh scenes, what the reference obtains from Mitsuba 3 + Dr.Jit:
 *   EPSMIntegrator.sample_path (Primal, log_path=True)   epsm.py:503-742   (records: :547, :648-654)
 *   ADIntegrator.prepare / sample_rays                    common.py:291-480 (PCG32 + TEA seeding:
 *                                                         src/render/sampler.cpp:115-134)
 *   PerspectiveCamera::sample_ray_differential            src/sensors/perspective.cpp:238-279
 *   Mesh::compute_surface_interaction (EPSM fields)       src/render/mesh.cpp:632-892
 *   diffuse / conductor / roughconductor / dielectric     src/bsdfs/{...}.cpp (+ the fork's BSDFSample3::hf,
 *                                                         roughconductor.cpp:255, and its forced
 *                                                         non-visible microfacet sampling, microfacet.h `if (true)`)
 *   area / point emitters, Scene::sample_emitter_direction src/emitters/{area,point}.cpp, src/render/scene.cpp:226-300
 *   film splat (box / gaussian) for the primal image      src/render/imageblock.cpp, src/rfilters/gaussian.cpp
 *
 * One lane = one path; bounces run in lock step inside ONE launch (no per-bounce host sync,
 * no .torch() round trips); every logged vertex is written in the record layout of epsm.h
 * together with the parameter addressing (EpsmScatterRecord), so the reference's second,
 * Backward-mode trace is not needed.
 *
 * PARITY UNPINNED: Mitsuba/Dr.Jit can be neither built nor imported here and the reference
 * has no test for these functions; they are pinned by analytic
 * known-answer tests (tests/test_tracer_*.py).
 */
#ifndef EPSM_TRACE_H
#define EPSM_TRACE_H

#include <stdint.h>
#include "epsm.h"

#ifdef __cplusplus
extern "C" {
#endif

/* mesh flags (the low four match EPSM_MODE_*) */
#define EPSM_MESH_VERTEX_NORMALS 0x1u
#define EPSM_MESH_FLIP_NORMALS   0x2u
#define EPSM_MESH_POS_ATTACHED   0x4u
#define EPSM_MESH_NRM_ATTACHED   0x8u
#define EPSM_MESH_IS_MESH        0x10u   /* 0: tessellated analytic shape (rectangle): si.ismesh stays 0 */
#define EPSM_MESH_HAS_UV         0x20u   /* EpsmScene.texcoords holds its vertices' (u, v); otherwise si.uv = (b1, b2) (mesh.cpp:736-745) */

enum { EPSM_BSDF_DIFFUSE_T = 0, EPSM_BSDF_CONDUCTOR_T = 1, EPSM_BSDF_ROUGHCONDUCTOR_T = 2, EPSM_BSDF_DIELECTRIC_T = 3 };
enum { EPSM_DISTR_BECKMANN = 0, EPSM_DISTR_GGX = 1 };
enum { EPSM_EMITTER_AREA = 0, EPSM_EMITTER_POINT = 1, EPSM_EMITTER_CONSTANT = 2, EPSM_EMITTER_ENVMAP = 3 };
enum { EPSM_RFILTER_BOX = 0, EPSM_RFILTER_GAUSSIAN = 1 };

/* Tracer flags.
 * EPSM_TRACE_SPARSE_LOG: for a bounce a path did NOT reach, only the four fields the gradient kernels' masks read
 *   (bsdf = 0, active = active_em = ismesh = 0) are written; every other array keeps its previous contents at that
 *   (path, bounce).  The reference logs masked lanes as zeros (epsm.py:551, 648-654) and so does the default; the
 *   zeros are 203 B per dead (path, bounce) that epsm_manifold_grad / _scatter / epsm_backward_pass never read. */
#define EPSM_TRACE_SPARSE_LOG 0x1u
/* EPSM_TRACE_PACKED_LOG: the vertex log is written in the NATIVE layout of the backward kernel (include/epsm.h,
 * EpsmPackedLog) instead of the per-field arrays: recs[0].packed = (N, K_log, 32) words, one 128-byte record per
 * (path, bounce) written with eight 16-byte stores, recs[0].pflags = (N) flag words (5 bits per bounce); bounces a path
 * did not reach are not touched (the flag word says so).  ray_o must then point to an (N,12) array that receives
 * o, d, d_x, d_y of a path side by side (ray_d / ray_dx / ray_dy are ignored); only `shadow` of the per-field pointers is
 * still used.  recs[0].ray_stride / packed_stride (words per path, multiples of 4) place both in one interleaved block per path.  The alpha slot of a vertex's BSDF is NOT in the record: the consumer takes it from the triangle table
 * (bits 8.. of the mode word). */
#define EPSM_TRACE_PACKED_LOG 0x2u
/* EPSM_TRACE_GRADIENT_ONLY: the trace feeds calc_grad and nothing else (render_backward's 5-channel branch, epsm.py:235-297:
 * the image of that pass is never used, :729-732) -- a path is RETIRED after the bounce that logs vertex k as soon as no
 * later vertex can produce, or be read by, a term of calc_grad, i.e. when the masks of epsm.py:793-803, 852-856, 916-921
 * (`valid`: every vertex so far is a mesh hit; `hasdiffuse == 0`) can no longer hold for any id >= k:
 *     manifold           goes on while every vertex so far is a mesh hit and none is Diffuse
 *     manifold_caustic   (with EPSM_TRACE_GRADIENT_CAUSTIC) goes on while vertex 1 is Diffuse, every vertex so far is a
 *                        mesh hit and fewer than two are Diffuse (epsm.py:998-999, 1172-1183)
 * and after vertex K_log in any case.  The visibility ray of an emitter sample is traced only where its answer is read: it
 * decides the logged weight eweight = sum Lr_dir, which calc_grad uses in the light-sampling term of a vertex behind which a
 * `manifold` path goes on (epsm.py:622-627, 852-855) and nowhere in manifold_caustic (its light terms are identically zero);
 * the occluder record of vertex 1 (max_depth <= 3, epsm.py:609-620) is traced as always.  With the native log a vertex that
 * retires its path by the rule -- a chain's end point, a diffuse first hit: only ever LOOKED at -- gets the first sector of its
 * record and its Diffuse / Null / active / mesh bits, nothing else (no emitter or BSDF sample is drawn for it; its active_em
 * bit stays 0).  Every word calc_grad reads is the one the full trace writes, so the gradients are identical; `radiance`,
 * the eweight words nobody reads and the second sector / active_em bit of such vertices are NOT those of the full trace: pass
 * radiance = valid = film_pos = NULL (with the native log nothing is then written for them and the finishing pass is skipped).  bench.py's real_scene leg: trace + log 14.3 -> see DESIGN.md 5b. */
#define EPSM_TRACE_GRADIENT_ONLY    0x4u
#define EPSM_TRACE_GRADIENT_CAUSTIC 0x8u
/* EPSM_TRACE_NO_TAIL (epsm_trace_paths_wavefront only; diagnostics): keep the three stages for EVERY bounce.  Without it the
 * paths still alive into a bounce >= 1 are carried through the rest of their loop by ONE launch as soon as fewer than 2^19 of
 * them are left (a stage cannot take less than one traversal's chain of cache misses, ~0.1 ms, however short its queue) --
 * same arithmetic per path, same results; the queue-length counters of the bounces behind that point then stay 0. */
#define EPSM_TRACE_NO_TAIL          0x10u
/* EPSM_TRACE_FUSE_FIRST_HIT (with EPSM_TRACE_GRADIENT_ONLY + EPSM_TRACE_PACKED_LOG, wavefront form only; recs[0].first_hit names the
 * backward pass's inputs): what epsm_backward_pass_packed would do for a path WITHOUT a chain is done by the stage that finds its
 * first hit (the primary rays' packet stage; the shade stage in EPSM_WF_NO_PACKET builds), and nothing of such a path is logged --
 *   - every path's share of d loss / d ray.o = -sum grad_d (epsm.py:255-261) goes into first_hit->grad_o_sum (when not NULL);
 *   - a path the rule retires at its first vertex (a diffuse, non-mesh or missed first hit: 94 % of the paths of the clutter scene)
 *     gives its first-vertex tangent's rows -- si_follow.p * diffuse_grad[0] = clamp(dldp) b_j into the hit triangle's vertex rows of
 *     first_hit->grad_pos (epsm.py:250-272, 561-562, 791-792) -- there, summed over the wave first, and its flag word is written as 0:
 *     the backward kernel, called with grad_o_sum = NULL for this log, gives it no lane and reads nothing of it.
 * Same sums as the unfused pair of calls (float order aside).  Not with the occluder record (max_depth <= 3). */
#define EPSM_TRACE_FUSE_FIRST_HIT   0x20u

typedef struct EpsmMesh {
    uint32_t tri_begin, tri_count;   /* this mesh's range in the triangle arrays */
    uint32_t flags;                  /* EPSM_MESH_* */
    int32_t bsdf;                    /* index into bsdfs */
    int32_t emitter;                 /* index into emitters, -1 = none */
    float area;                      /* total surface area (emitter sampling pdf) */
    uint32_t cdf_begin;              /* first entry of this mesh's triangle-area CDF in emitter_cdf (tri_count entries) */
    uint32_t pad;
} EpsmMesh;

typedef struct EpsmBsdf {
    uint32_t type;                   /* EPSM_BSDF_*_T */
    uint32_t twosided;
    uint32_t distr;                  /* EPSM_DISTR_* (roughconductor) */
    uint32_t sample_visible;         /* roughconductor: selects the weight/pdf formulas (roughconductor.cpp:258-262) */
    float reflectance[3];            /* diffuse reflectance / specular_reflectance */
    float alpha;                     /* roughconductor roughness */
    float eta[3], k[3];              /* conductor complex IOR; eta = 0,k = 1 is the '100 % reflecting mirror' */
    float int_ior, ext_ior;          /* dielectric */
    int32_t alpha_slot;              /* slot of this BSDF's alpha in grad_alpha, -1 = not optimised */
    int32_t color_slot;              /* diffuse: slot of `reflectance` in the colour adjoint of epsm_trace_paths_color,
                                        -1 = not optimised */
    int32_t texture;                 /* diffuse: index into EpsmScene.textures of the `bitmap` its reflectance is, -1 = `reflectance` */
    uint32_t pad;
} EpsmBsdf;

/* A `bitmap` texture (src/textures/bitmap.cpp): linear RGB texels, looked up at si.uv as there -- uv * (width, height) - 0.5,
 * bilinear between the four texels around it (or the nearest one), indices wrapped (`repeat`). */
typedef struct EpsmTexture {
    const float *texels;             /* (height, width, 3), row 0 at v = 0 */
    int32_t width, height;
    uint32_t nearest;                /* filter_type: 0 bilinear, 1 nearest */
    uint32_t pad;
} EpsmTexture;

typedef struct EpsmEmitter {
    uint32_t type;                   /* EPSM_EMITTER_* */
    int32_t mesh;                    /* area: emitting mesh */
    float radiance[3];               /* area, constant: radiance; point: intensity; envmap: unused (EpsmEnvironment.texels) */
    float position[3];               /* point */
    int32_t color_slot;              /* slot of `radiance` in the colour adjoint, -1 = not optimised */
    uint32_t pad;
} EpsmEmitter;

typedef struct EpsmBvhNode {         /* 128 bytes: the boxes of up to FOUR children, one component of all four per 16-byte quad */
    float lox[4], loy[4], loz[4];    /* lower corners (absent child: +inf) */
    float hix[4], hiy[4], hiz[4];    /* upper corners (absent child: -inf) */
    int32_t c[4];                    /* child reference: >= 0 node index; < 0 leaf ~((first << 3) | count) with `first`
                                        the leaf's first triangle in tri_verts / prim_index and count <= 7;
                                        0x7fffffff = absent */
    int32_t n[4];                    /* build-side copy of the leaf counts (0 = inner); not read by the traversal */
} EpsmBvhNode;

typedef struct EpsmSensor {
    float to_world[12];              /* 3x4 row-major camera-to-world (rotation | translation) */
    float sample_to_camera[16];      /* 4x4 row-major, perspective.cpp:171-176 */
    float dx[3], dy[3];              /* position differentials on the near plane, perspective.cpp:178-182 */
    float near_clip, far_clip;
    int32_t width, height;
    int32_t border;                  /* film.sample_border: samples are also generated for `border` pixels around the film
                                        (rfilter.border_size(): 2 for the gaussian, 0 for the box), common.py:309-336; the
                                        wavefront is then (width + 2 border) (height + 2 border) spp paths */
    int32_t pad;
} EpsmSensor;

/* The scene's environment emitter (src/emitters/constant.cpp, envmap.cpp): at most one, `kind` says which (0: none) and
 * `emitter` = its index in EpsmScene.emitters.  Every tracer entry point checks: kind in {0, 1, 2}; kind != 0 => 0 <= emitter <
 * n_emitters; kind == ENVMAP => the four tables non-NULL and width, height >= 2; n_textures > 0 => textures non-NULL (EPSM_EINVAL).  A ray that leaves the scene sees its radiance (MIS against emitter sampling as for an area
 * light); an emitter sample is the point ref + 2 max(radius, |ref - center|) d (constant.cpp:113-116, envmap.cpp:398-399) -- what
 * the vertex log records as `light` -- with d uniform on the sphere (constant) or drawn from the map.
 * envmap: `texels` is the (height, width + 1, 3) lat-long map, `scale` applied, column `width` a copy of column 0; texel (i, j)
 * sits at phi = 2 pi (i + 1/2) / width, theta = pi j / (height - 1) (envmap.cpp:387-395, 416-422), direction
 * (sin phi sin theta, cos theta, -cos phi sin theta) in the emitter's frame; radiance is bilinear in between.  Sampling is by
 * CELL (the width x (height - 1) bilinear patches): weight = mean over the four corners of luminance x sin theta, rows by
 * `row_cdf`, columns by `col_cdf`, uniform inside a cell -- a piecewise-constant stand-in for the reference's hierarchical
 * sample warp (same support, same estimator up to variance). */
enum { EPSM_ENV_NONE = 0, EPSM_ENV_CONSTANT = 1, EPSM_ENV_ENVMAP = 2 };
typedef struct EpsmEnvironment {
    int32_t kind;                    /* EPSM_ENV_*: 0 = the scene has no environment emitter, so a zero-initialised EpsmScene is
                                        safe (ABI 6; until ABI 5 `emitter = -1` said so and 0 named emitter 0).  `kind`, not
                                        emitters[emitter].type, decides which tables the device code touches. */
    int32_t emitter;                 /* kind != 0: its index in EpsmScene.emitters (radiance, colour slot, emitter choice) */
    int32_t width, height;           /* envmap only */
    const float *texels;             /* (height, width + 1, 3) */
    const float *row_cdf;            /* (height - 1) cumulative, normalised */
    const float *col_cdf;            /* (height - 1, width) cumulative per row, normalised */
    const float *cell_pdf;           /* (height - 1, width) density of a cell in (u, v) in [0,1)^2 */
    float to_local[9];               /* world -> emitter frame, row-major rotation */
    float center[3], radius;         /* bounding sphere of shapes and sensors */
} EpsmEnvironment;

typedef struct EpsmScene {           /* host struct holding DEVICE pointers */
    const float *positions;          /* (V,3) world space */
    const float *normals;            /* (V,3) (zero rows for meshes without vertex normals) */
    const uint32_t *tri;             /* (T,3) rows of the triangle's vertices in positions/normals, mesh by mesh */
    const uint32_t *tri_mesh;        /* (T)   owning mesh */
    const EpsmMesh *meshes;          int32_t n_meshes;
    const EpsmBsdf *bsdfs;           int32_t n_bsdfs;
    const EpsmEmitter *emitters;     int32_t n_emitters;
    const float *emitter_cdf;        /* concatenated normalised area CDFs of the emitting meshes */
    const EpsmBvhNode *bvh;          int32_t n_nodes;   /* node 0 is the root; depth <= 16 (four-wide) */
    const uint32_t *prim_index;      /* leaf entries: BVH order -> triangle id (triangles stay mesh-contiguous) */
    const float *tri_verts;          /* (T,9) p0,p1,p2 of the triangles in BVH (leaf) order */
    int64_t n_vertices, n_triangles;
    EpsmEnvironment env;
    const float *texcoords;          /* (V,2) per-vertex (u, v) of the meshes flagged EPSM_MESH_HAS_UV (zero rows elsewhere), or NULL */
    const EpsmTexture *textures;     int32_t n_textures;
} EpsmScene;

/* EPSM_TRACE_FUSE_FIRST_HIT: the arguments epsm_backward_pass_packed takes for the same tile (include/epsm.h) */
typedef struct EpsmFirstHitBackward {
    const float *grad_img;           /* (.., img_width, img_channels): channels 3, 4 = d loss / d film position */
    int img_width, img_channels, res;
    float clip;                      /* outlier clamp of calc_grad (0.1 in the reference); <= 0 or non-finite: none */
    const uint32_t *tri_table;       /* (T,4) [v0, v1, v2, mode] */
    int64_t T, V;
    float *grad_pos;                 /* (V,3), accumulated */
    float *grad_o_sum;               /* (3), accumulated; may be NULL */
    uint32_t *survivors;             /* (N) or NULL: receives, in path order, the indices of the paths this stage did NOT retire -- the only
                                        ones of which the log holds anything -- and *survivor_count their number: EpsmPackedLog.path_list /
                                        path_count of the backward pass that follows (its windows then run over these paths alone) */
    uint32_t *survivor_count;        /* (1) or NULL (both or neither) */
} EpsmFirstHitBackward;

/* Writable twin of EpsmVertexRecord + EpsmScatterRecord for one logged bounce. */
typedef struct EpsmRecordOut {
    float *p0, *p1, *p2, *p;         /* (N,3) */
    float *n0, *n1, *n2, *normal;    /* (N,3) */
    float *b0, *b1, *eta;            /* (N) */
    float *hf, *light;               /* (N,3) */
    uint32_t *bsdf;                  /* (N) */
    uint8_t *active, *active_em, *ismesh;  /* (N) */
    uint32_t *tri, *aux, *emit;      /* (N) (N,4) (N,4): EpsmScatterRecord (tri = triangle id, emit = [etri, eb0, eb1, ew]) */
    float *packed;                   /* EPSM_TRACE_PACKED_LOG, recs[0] only: (N, K_log, 32) words */
    uint32_t *pflags;                /* EPSM_TRACE_PACKED_LOG, recs[0] only: (N) flag words */
    uint32_t *shadow;                /* (N,4): EpsmScatterRecord.shadow [stri, sb0, sb1, dis]; written for the first logged vertex only and
                                        only meaningful when max_depth <= 3 (epsm.py:610); may be NULL */
    int64_t ray_stride, packed_stride; /* EPSM_TRACE_PACKED_LOG, recs[0] only (ABI v7): words between the rays (at ray_o) / the first
                                        records (at packed) of consecutive paths; 0 = dense, 12 and 32 K_log.  The interleaved block
                                        of include/epsm.h (EpsmPackedLog): ray_o = block, packed = block + 16, both 32 (K_log + 1) */
    const EpsmFirstHitBackward *first_hit; /* EPSM_TRACE_FUSE_FIRST_HIT, recs[0] only (a HOST pointer: read during the call) */
} EpsmRecordOut;

/* ---------------------------------------------------------------------------
 * epsm_trace_paths -- sample_rays + sample_path(Primal, log_path=True) for paths
 *   [path_offset, path_offset + N) of the wavefront  width*height*spp  (ordered pixel-major,
 *   then sample: common.py:320-330).
 *   seed, spp, max_depth, rr_depth   as in ADIntegrator (common.py:28-43, 424-480)
 *   K_log                            bounces to log (<= 5, epsm.py:648)
 *   ray_o/d/dx/dy (N,3), film_pos (N,2), radiance (N,3), valid (N) u8: outputs (any may be NULL
 *                                    except ray_*); radiance = L of epsm.py:658, valid = depth != 0
 *   recs                             K_log records to fill (all fields written for every path)
 *   flags                            0 or any of EPSM_TRACE_SPARSE_LOG, EPSM_TRACE_PACKED_LOG, EPSM_TRACE_GRADIENT_ONLY (+ _CAUSTIC)
 * ------------------------------------------------------------------------- */
int epsm_trace_paths(const EpsmScene *scene, const EpsmSensor *sensor,
                     uint32_t seed, int spp, int max_depth, int rr_depth,
                     int64_t path_offset, int64_t N, int K_log,
                     float *ray_o, float *ray_d, float *ray_dx, float *ray_dy,
                     float *film_pos, float *radiance, uint8_t *valid,
                     const EpsmRecordOut *recs, uint32_t flags, void *stream);

/* ---------------------------------------------------------------------------
 * epsm_trace_paths_wavefront -- the same function (same arguments, same per-path results) run as a
 *   wavefront of stages with compaction between the bounces: per bounce one closest-hit kernel, one
 *   shading kernel (the loop body of epsm.py:551-735) and one shadow-ray kernel over QUEUES of the
 *   paths that are still alive, then one pass that writes radiance / valid and the inactive-zero
 *   records of the bounces a path never reached.  Pays on scenes with many triangles, where the
 *   one-launch form is bound by divergence (dead lanes, mixed closest-hit / shadow traversals) and by
 *   the occupancy its register count allows; costs 172 B/path/bounce of state traffic, so small scenes
 *   are faster in one launch.
 *   workspace        device memory, 16-byte aligned, >= epsm_trace_workspace_bytes(N); contents are
 *                    scratch (no state is kept between calls)
 * ------------------------------------------------------------------------- */
size_t epsm_trace_workspace_bytes(int64_t N);
int epsm_trace_paths_wavefront(const EpsmScene *scene, const EpsmSensor *sensor,
                               uint32_t seed, int spp, int max_depth, int rr_depth,
                               int64_t path_offset, int64_t N, int K_log,
                               float *ray_o, float *ray_d, float *ray_dx, float *ray_dy,
                               float *film_pos, float *radiance, uint8_t *valid,
                               const EpsmRecordOut *recs, uint32_t flags, void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------
 * epsm_trace_paths_color -- epsm_trace_paths (no vertex log) that also returns, per path, the derivative of its
 *   radiance w.r.t. the COLOUR parameters attached to the scene (EpsmBsdf.color_slot: diffuse reflectance;
 *   EpsmEmitter.color_slot: emitted radiance / intensity) -- the colour adjoint of the reference's hybrid phase
 *   (3-channel grad_in: epsm.py:230-234 -> prb-style backward, src/python/python/ad/integrators/prb.py) for parameters
 *   the radiance is multiplicative in.  Sampling is detached as in PRB: a term T of the estimator that passed n_j
 *   vertices of BSDF j satisfies dT/d rho_j,c = n_j T_c / rho_j,c, and dT/dE_e,c = T_c / E_e,c for the emitter it ends on.
 *     color_sum   (N, n_color, 3) f32, written: sum over the path's terms of n_j T_c (BSDF slots) / T_c (emitter
 *                 slots); the caller divides by the parameter value and contracts with the adjoint radiance
 *     n_color     number of slots, <= 4
 *   One-launch form only.  Visibility / geometry derivatives (prb_reparam's warp field) are NOT part of it.
 * ------------------------------------------------------------------------- */
int epsm_trace_paths_color(const EpsmScene *scene, const EpsmSensor *sensor,
                           uint32_t seed, int spp, int max_depth, int rr_depth,
                           int64_t path_offset, int64_t N,
                           float *film_pos, float *radiance, uint8_t *valid,
                           float *color_sum, int n_color, void *stream);

/* ---------------------------------------------------------------------------
 * epsm_trace_paths_reparam -- the second pass of RBIntegrator.render_backward for `prb_reparam`
 *   (src/python/python/ad/integrators/common.py:944-955, prb_reparam.py:277-607, ad/reparam.py:10-333): paths
 *   [path_offset, path_offset + N) are replayed under the primal pass's seed and the gradient of
 *   sum(image * grad_in) w.r.t. the VERTEX POSITIONS (and vertex normals) of the meshes flagged EPSM_MESH_POS_ATTACHED
 *   (EPSM_MESH_NRM_ATTACHED) is ACCUMULATED into grad_pos / grad_nrm -- through shading, and through visibility by the
 *   warp field of Bangaru et al. (auxiliary rays, harmonic weights, divergence).
 *     radiance      (N,3) L of every path from the primal pass (epsm_trace_paths with the same seed / spp / depth)
 *     adj_radiance  (N,3) d loss / d L               } the adjoint of splat + weight division, which the caller owns
 *     adj_film      (N,3) d loss / d film position (x, y in pixels) and d loss / d det of the primary ray's
 *                         reparameterisation (common.py:405-418, 888-903)
 *     reparam_max_depth, reparam_rays (<= 64), kappa, exponent   prb_reparam.py:226-250
 *     flags         EPSM_REPARAM_ANTITHETIC: auxiliary rays 2m and 2m + 1 of a warp share one sample, the even one mirrored
 *                   about the ray (`reparam_antithetic`, prb_reparam.py:243-246, reparam.py:82-84, 189-196)
 *     grad_pos, grad_nrm   (V,3) f32 device buffers, float atomics; grad_nrm may be NULL
 *     workspace            device memory, 16-byte aligned, >= epsm_trace_reparam_workspace_bytes(N); scratch
 *   Two launches: (1) a lane replays its path, differentiates each vertex (dual numbers) and leaves one 64-byte REQUEST
 *   per reparameterize_ray call -- ray, adjoint of its direction and divergence, where its origin is glued -- at most
 *   13 per path; (2) one lane per AUXILIARY RAY: the 16 / 32 / 64 lanes of a request trace its rays side by side, reduce
 *   the weights over the group and scatter the warp field's adjoint.
 * ------------------------------------------------------------------------- */
#define EPSM_REPARAM_ANTITHETIC 1u
size_t epsm_trace_reparam_workspace_bytes(int64_t N);
int epsm_trace_paths_reparam(const EpsmScene *scene, const EpsmSensor *sensor,
                             uint32_t seed, int spp, int max_depth, int rr_depth,
                             int64_t path_offset, int64_t N,
                             const float *radiance, const float *adj_radiance, const float *adj_film,
                             int reparam_max_depth, int reparam_rays, float kappa, float exponent, uint32_t flags,
                             float *grad_pos, float *grad_nrm, void *workspace, size_t workspace_bytes, void *stream);

/* epsm_film_splat -- ImageBlock::put + weight division (film.develop): accumulates
 * radiance with the reconstruction filter into accum (height,width,4) [r,g,b,w] (atomics);
 * epsm_film_develop divides into image (height,width,3). */
int epsm_film_splat(int64_t N, const float *film_pos, const float *radiance, int width, int height,
                    int rfilter, float *accum, void *stream);
int epsm_film_develop(int width, int height, const float *accum, float *image, void *stream);

/* epsm_film_adjoint_reparam -- the adjoint of splat + weight division between the two passes of prb_reparam's render_backward
 * (common.py:880-920: image[p] = sum_i w_ip L_i det_i / sum_i w_ip det_i with the gaussian reconstruction filter, radius 2):
 * per sample i of the primal pass, from its film position (N,2), its radiance (N,3), the gradient image grad_img
 * (height,width,grad_channels >= 3; the first three channels are read) and the primal film accum (height,width,4) [r,g,b,w]:
 *   dL  (N,3)  d loss / d radiance_i
 *   adj (N,3)  d loss / d film_pos_i.x, d loss / d film_pos_i.y, d loss / d det_i   (at det = 1)
 * which epsm_trace_paths_reparam takes as adj_radiance / adj_film.  One kernel, no allocation. */
int epsm_film_adjoint_reparam(int64_t N, const float *film_pos, const float *radiance, const float *grad_img, int grad_channels,
                              const float *accum, int width, int height, float *dL, float *adj, void *stream);

/* epsm_probe -- evaluates ONE of the tracer's per-path functions on n rows of plain numbers, on the device, with the very
 * code the tracer runs (csrc/epsm_probe_core.h).  It exists so that the known answers the reference's own unit tests hold
 * for these functions can be checked against the product (tests/golden/reference_vectors.py):
 *   EPSM_PROBE_TEA                in v0, v1 (u32 bits)                          out v0', v1' (bits)     include/mitsuba/core/random.h:77-104  (src/core/tests/test_random.py:9-27)
 *   EPSM_PROBE_PCG32              in initstate lo, hi, initseq lo, hi (bits)    out 6 draws (bits), then 6 x next_1d()   drjit PCG32::seed / next (src/samplers/tests/test_independent.py:16-28)
 *   EPSM_PROBE_SAMPLER            in seed, wavefront index (bits)               out 12 x next_1d() of that path's stream    src/render/sampler.cpp:115-134
 *   EPSM_PROBE_MICROFACET         in m (3), wi (3); cfg = EpsmBsdf (distr, alpha, sample_visible)    out D(m), pdf(wi, m), smith_g1(m, wi)    include/mitsuba/render/microfacet.h (src/render/tests/test_microfacet.py)
 *   EPSM_PROBE_MICROFACET_SAMPLE  in u1, u2; cfg = EpsmBsdf                     out m (3), pdf, d m / d alpha (3)
 *   EPSM_PROBE_FRESNEL            in cos_theta_i, eta                           out F, cos_theta_t, eta_it, eta_ti          include/mitsuba/render/fresnel.h:34-72 (src/render/tests/test_fresnel.py)
 *   EPSM_PROBE_FRESNEL_CONDUCTOR  in cos_theta_i, eta, k                        out F                                        fresnel.h:92-117
 *   EPSM_PROBE_RFILTER            in x                                          out gaussian reconstruction filter at x      src/rfilters/gaussian.cpp (src/rfilters/tests/test_rfilter.py:14-19)
 *   EPSM_PROBE_PRIMARY_RAY        in film position (pixels); cfg = EpsmSensor   out o, d, d_x, d_y (3 each)                  src/sensors/perspective.cpp:238-279 (src/sensors/tests/test_perspective.py:89-135)
 *   EPSM_PROBE_BSDF_SAMPLE        in wi (3), sample1, sample2 (2); cfg = EpsmBsdf out wo (3), weight (3), pdf, eta, sampled_type (bits), valid   src/bsdfs/{diffuse,conductor,roughconductor,dielectric,twosided}.cpp sample() (src/bsdfs/tests/test_dielectric.py:31-158)
 *   EPSM_PROBE_BSDF_EVAL          in wi (3), wo (3); cfg = EpsmBsdf             out value incl. cosine (3), pdf              eval_pdf() (src/bsdfs/tests/test_diffuse.py:13-35, test_twosided.py:29-45)
 * in: (n, EPSM_PROBE_IN) floats, out: (n, EPSM_PROBE_OUT) floats, device pointers; cfg: HOST pointer to the struct named
 * above (NULL otherwise).  Not on any hot path. */
enum { EPSM_PROBE_TEA = 0, EPSM_PROBE_PCG32 = 1, EPSM_PROBE_SAMPLER = 2, EPSM_PROBE_MICROFACET = 3, EPSM_PROBE_MICROFACET_SAMPLE = 4,
       EPSM_PROBE_FRESNEL = 5, EPSM_PROBE_FRESNEL_CONDUCTOR = 6, EPSM_PROBE_RFILTER = 7, EPSM_PROBE_PRIMARY_RAY = 8, EPSM_PROBE_BSDF_SAMPLE = 9,
       EPSM_PROBE_BSDF_EVAL = 10, EPSM_PROBE_COUNT = 11 };
#define EPSM_PROBE_IN 8
#define EPSM_PROBE_OUT 16
int epsm_probe(int what, int64_t n, const float *in, float *out, const void *cfg, void *stream);

#ifdef __cplusplus
}
#endif
#endif
