/*
 * epsm_oracle_aux.c -- TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * CPU restatements (float64; OpenMP over paths, so sums are deterministic up to the order of float64 additions --
 * run with OMP_NUM_THREADS=1 for bit-reproducible results) of the two pieces of
 * EPSMIntegrator.render_backward that surround calc_grad:
 *
 *   epsm_oracle_first_vertex_tangent   epsm.py:238-272 with the Moeller-Trumbore
 *       intersection of include/mitsuba/render/mesh.h:343-365 and the barycentric
 *       mapping of src/render/mesh.cpp:698-709.  The reference gets the tangent by
 *       forward-mode AD (dr.set_grad(ray.d), dr.forward_to(si.p)); this file does
 *       the same thing literally, with dual numbers, rather than with the closed
 *       form the HIP kernel uses.
 *   epsm_oracle_scatter                epsm.py:559-562, 622-627, 644-645 + the gather
 *       adjoints of include/mitsuba/render/mesh.h:94-106 and the shading-normal code
 *       of src/render/mesh.cpp:729,784-790,811-827.
 *
 * PINNED BY REFERENCE-HELD KNOWN ANSWERS: these parts of the reference need Dr.Jit/Mitsuba, which
 * can be neither built nor imported here, but its own unit tests hold known answers for exactly
 * these derivatives -- src/render/tests/test_mesh.py:380-455 (d p, d uv, d t under d ray.o / d ray.d
 * and the adjoints to ray.o) and :560-640 (gather adjoints of si.p, si.n, si.sh_frame.n into
 * vertex_positions) -- typed as VALUES into tests/golden/reference_vectors.py and checked by
 * tests/test_reference_vectors.py.  In addition: finite differences of the float64 ray/triangle
 * intersection and torch autograd (float64) of the reference's own loss expressions
 * (tests/test_tangent_scatter_oracle.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/epsm.h"

typedef struct { double v, d; } dual;
static inline dual dmk(double v, double d) { dual r = {v, d}; return r; }
static inline dual dadd(dual a, dual b) { return dmk(a.v + b.v, a.d + b.d); }
static inline dual dsub(dual a, dual b) { return dmk(a.v - b.v, a.d - b.d); }
static inline dual dmul(dual a, dual b) { return dmk(a.v * b.v, a.d * b.v + a.v * b.d); }
static inline dual drcp(dual a) { return dmk(1.0 / a.v, -a.d / (a.v * a.v)); }
typedef struct { dual x, y, z; } dvec;
static inline dvec dvsub(dvec a, dvec b) { dvec r = {dsub(a.x, b.x), dsub(a.y, b.y), dsub(a.z, b.z)}; return r; }
static inline dual dvdot(dvec a, dvec b) { return dadd(dadd(dmul(a.x, b.x), dmul(a.y, b.y)), dmul(a.z, b.z)); }
static inline dvec dvcross(dvec a, dvec b) {
    dvec r = {dsub(dmul(a.y, b.z), dmul(a.z, b.y)), dsub(dmul(a.z, b.x), dmul(a.x, b.z)),
              dsub(dmul(a.x, b.y), dmul(a.y, b.x))};
    return r;
}
static inline dvec dconst(const float *p) { dvec r = {dmk(p[0], 0), dmk(p[1], 0), dmk(p[2], 0)}; return r; }

/* Moeller-Trumbore (include/mitsuba/render/mesh.h:349-362) in dual numbers: t, u, v and their tangents when the ray's origin
 * and direction move */
static void mt_dual(dvec o, dvec d, dvec q0, dvec q1, dvec q2, dual *t, dual *u, dual *v) {
    dvec e1 = dvsub(q1, q0), e2 = dvsub(q2, q0);
    dvec pvec = dvcross(d, e2);
    dual inv_det = drcp(dvdot(e1, pvec));
    dvec tvec = dvsub(o, q0);
    *u = dmul(dvdot(tvec, pvec), inv_det);
    dvec qvec = dvcross(tvec, e1);
    *v = dmul(dvdot(d, qvec), inv_det);
    *t = dmul(dvdot(e2, qvec), inv_det);
}

/* One ray, one triangle, forward mode: the origin moves with velocity odot, the direction with ddot (what
 * dr.forward(ray.o.x) / dr.forward(ray.d.x) do in src/render/tests/test_mesh.py:380-421).
 * out = [t, u, v,  dt, du, dv,  dp (3)]  with  si.p = p0 b0 + p1 b1 + p2 b2,  b1 = u, b2 = v, b0 = 1 - u - v (mesh.cpp:698-709). */
int epsm_oracle_intersect_tangent(const double *o, const double *d, const double *odot, const double *ddot,
                                  const double *p0, const double *p1, const double *p2, double *out) {
    dvec O = {dmk(o[0], odot[0]), dmk(o[1], odot[1]), dmk(o[2], odot[2])};
    dvec D = {dmk(d[0], ddot[0]), dmk(d[1], ddot[1]), dmk(d[2], ddot[2])};
    dvec q0 = {dmk(p0[0], 0), dmk(p0[1], 0), dmk(p0[2], 0)}, q1 = {dmk(p1[0], 0), dmk(p1[1], 0), dmk(p1[2], 0)},
         q2 = {dmk(p2[0], 0), dmk(p2[1], 0), dmk(p2[2], 0)};
    dual t, u, v;
    mt_dual(O, D, q0, q1, q2, &t, &u, &v);
    out[0] = t.v; out[1] = u.v; out[2] = v.v; out[3] = t.d; out[4] = u.d; out[5] = v.d;
    const double db0 = -u.d - v.d;
    for (int c = 0; c < 3; ++c) out[6 + c] = p0[c] * db0 + p1[c] * u.d + p2[c] * v.d;
    return 0;
}

int epsm_oracle_first_vertex_tangent(int64_t N, int64_t path_offset, int spp, int res,
                                     const float *ray_o, const float *ray_d,
                                     const float *ray_dx, const float *ray_dy,
                                     const float *grad_img, int img_width, int img_channels,
                                     const float *p0, const float *p1, const float *p2,
                                     const uint8_t *active,
                                     double *dlduv, int64_t dlduv_stride, double *dldp,
                                     double *grad_o_sum) {
    double go0 = 0.0, go1 = 0.0, go2 = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : go0, go1, go2)
    for (int64_t i = 0; i < N; ++i) {
        int64_t pix = (path_offset + i) / spp, y = pix / res, x = pix % res;
        const float *g = grad_img + (y * img_width + x) * img_channels;
        double gx = g[3], gy = g[4];
        double gd[3];
        for (int c = 0; c < 3; ++c)   /* epsm.py:255 */
            gd[c] = ((double) ray_dx[3 * i + c] - ray_d[3 * i + c]) * gx + ((double) ray_dy[3 * i + c] - ray_d[3 * i + c]) * gy;
        go0 -= gd[0]; go1 -= gd[1]; go2 -= gd[2];   /* epsm.py:260-261 */
        double *row = dlduv + i * dlduv_stride;
        for (int64_t c = 0; c < dlduv_stride; ++c) row[c] = 0.0;
        dldp[3 * i] = dldp[3 * i + 1] = dldp[3 * i + 2] = 0.0;
        if (!active[i]) continue;
        /* ray.d carries the tangent grad_d (dr.set_grad(ray.d, grad_d), epsm.py:264) */
        dvec d = {dmk(ray_d[3 * i], gd[0]), dmk(ray_d[3 * i + 1], gd[1]), dmk(ray_d[3 * i + 2], gd[2])};
        dvec o = dconst(ray_o + 3 * i), q0 = dconst(p0 + 3 * i), q1 = dconst(p1 + 3 * i), q2 = dconst(p2 + 3 * i);
        dual t, u, v;
        mt_dual(o, d, q0, q1, q2, &t, &u, &v);   /* mesh.h:349-362 */
        (void) t;
        /* mesh.cpp:698-709 */
        dual b1 = u, b2 = v, b0 = dsub(dsub(dmk(1, 0), b1), b2);
        row[0] = b0.d;   /* epsm.py:268 */
        row[1] = b1.d;   /* epsm.py:269 */
        for (int c = 0; c < 3; ++c)   /* si.p = p0 b0 + p1 b1 + p2 b2 ; epsm.py:270 */
            dldp[3 * i + c] = p0[3 * i + c] * b0.d + p1[3 * i + c] * b1.d + p2[3 * i + c] * b2.d;
    }
    if (grad_o_sum) { grad_o_sum[0] = go0; grad_o_sum[1] = go1; grad_o_sum[2] = go2; }
    return 0;
}

/* Sums go either straight into the caller's buffers with an atomic add (one thread, or buffers too large to copy
 * per thread) or into a PRIVATE copy per thread that is reduced at the end -- what a CPU implementation that wants to
 * be fast does when every path hits the same few emitter rows. */
static int g_private = 0;
static void add1(double *p, double x) {
    if (g_private) { *p += x; return; }
#pragma omp atomic
    *p += x;
}
static void add3(double *buf, uint32_t v, const double *g, double w) {
    add1(buf + 3 * (int64_t) v + 0, g[0] * w);
    add1(buf + 3 * (int64_t) v + 1, g[1] * w);
    add1(buf + 3 * (int64_t) v + 2, g[2] * w);
}
/* row `id` of the scene's triangle table (include/epsm.h): [v0, v1, v2, mode]; ids beyond the table address nothing */
static int table_row(const uint32_t *table, int64_t T, uint32_t id, uint32_t row[4]) {
    if ((int64_t) id >= T) { row[0] = row[1] = row[2] = 0xFFFFFFFFu; row[3] = 0u; return 0; }
    memcpy(row, table + 4 * (int64_t) id, 16);
    return 1;
}
static void cross(const double *a, const double *b, double *c) {
    c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
}

/* Inputs as epsm_scatter (include/epsm.h) with HOST pointers; record arrays are
 * fp32 (as logged), gradients in/out are fp64.  grad buffers are accumulated. */
int epsm_oracle_scatter(int variant, int64_t N, int K,
                        const EpsmVertexRecord *verts, const EpsmScatterRecord *sc,
                        const uint32_t *tri_table, int64_t T,
                        const double *out_param, const double *out_light, const double *out_diffuse,
                        double *grad_pos, double *grad_nrm, double *grad_alpha, int64_t V, int64_t B) {
    const int P = variant == EPSM_VARIANT_MANIFOLD_CAUSTIC ? 5 * K - 2 : 5 * K;
    int nthreads = 1;
#ifdef _OPENMP
    nthreads = omp_get_max_threads();
#endif
    const int64_t per_thread = 6 * V + (grad_alpha ? B : 0);
    double *priv = 0;
    g_private = 0;
    if (nthreads > 1 && per_thread * nthreads <= (int64_t) 1 << 28) {          /* <= 2 GiB of private copies */
        priv = (double *) calloc((size_t) (per_thread * nthreads), sizeof(double));
        g_private = priv != 0;
    }
    double *const out_pos = grad_pos, *const out_nrm = grad_nrm, *const out_alpha = grad_alpha;
#pragma omp parallel
    {
#ifdef _OPENMP
    const int tid = omp_get_thread_num();
#else
    const int tid = 0;
#endif
    double *grad_pos = priv ? priv + (int64_t) tid * per_thread : out_pos;                 /* shadow the arguments */
    double *grad_nrm = priv ? grad_pos + 3 * V : out_nrm;
    double *grad_alpha = priv ? (out_alpha ? grad_pos + 6 * V : 0) : out_alpha;
#pragma omp for schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        for (int it = 0; it < K; ++it) {
            const EpsmVertexRecord *v = &verts[it];
            const EpsmScatterRecord *s = &sc[it];
            uint32_t vi[4];
            table_row(tri_table, T, s->tri[i], vi);
            const uint32_t mode = vi[3];
            const int idx_ok = vi[0] < (uint64_t) V && vi[1] < (uint64_t) V && vi[2] < (uint64_t) V;
            const double b0 = ((const float *) v->b0)[i], b1 = ((const float *) v->b1)[i], b2 = 1.0 - b0 - b1;
            const double bw[3] = {b0, b1, b2};
            const int has_nm = it * 5 + 4 < P;   /* epsm.py:559,644 */
            /* epsm.py:559-560 */
            if (has_nm && idx_ok && (mode & EPSM_MODE_POS_ATTACHED))
                for (int j = 0; j < 3; ++j) add3(grad_pos, vi[j], out_param + ((int64_t) (5 * it + j) * N + i) * 3, 1.0);
            /* epsm.py:561-562: si_follow.p = sum_j b_j p_j with detached b */
            if (idx_ok && (mode & EPSM_MODE_POS_ATTACHED))
                for (int j = 0; j < 3; ++j) add3(grad_pos, vi[j], out_diffuse + ((int64_t) it * N + i) * 3, bw[j]);
            /* epsm.py:644-645 */
            if (has_nm) {
                const double *gn = out_param + ((int64_t) (5 * it + 3) * N + i) * 3;
                const double sgn = (mode & EPSM_MODE_FLIP_NORMALS) ? -1.0 : 1.0;
                if (mode & EPSM_MODE_VERTEX_NORMALS) {
                    if (idx_ok && (mode & EPSM_MODE_NRM_ATTACHED)) {
                        /* buffer normals m_j = sgn * logged n_j; sh = sgn * normalize(sum b_j m_j)  (mesh.cpp:784-790,820-827) */
                        double m[3] = {0, 0, 0};
                        const float *nj[3] = {(const float *) v->n0 + 3 * i, (const float *) v->n1 + 3 * i, (const float *) v->n2 + 3 * i};
                        for (int j = 0; j < 3; ++j) for (int c = 0; c < 3; ++c) m[c] += bw[j] * sgn * nj[j][c];
                        double len = sqrt(m[0] * m[0] + m[1] * m[1] + m[2] * m[2]);
                        double mh[3] = {m[0] / len, m[1] / len, m[2] / len};
                        double dotg = mh[0] * gn[0] + mh[1] * gn[1] + mh[2] * gn[2];
                        double mb[3];   /* adjoint of m for loss = sgn * mh . gn */
                        for (int c = 0; c < 3; ++c) mb[c] = sgn * (gn[c] - mh[c] * dotg) / len;
                        for (int j = 0; j < 3; ++j) add3(grad_nrm, vi[j], mb, bw[j]);
                    }
                } else if (idx_ok && (mode & EPSM_MODE_POS_ATTACHED)) {
                    /* flat: sh = sgn * normalize(cross(p1-p0, p2-p0))  (mesh.cpp:729,811,820-827) */
                    const float *q0 = (const float *) v->p0 + 3 * i, *q1 = (const float *) v->p1 + 3 * i, *q2 = (const float *) v->p2 + 3 * i;
                    double d0[3], d1[3], c[3];
                    for (int a = 0; a < 3; ++a) { d0[a] = (double) q1[a] - q0[a]; d1[a] = (double) q2[a] - q0[a]; }
                    cross(d0, d1, c);
                    double len = sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
                    double ch[3] = {c[0] / len, c[1] / len, c[2] / len};
                    double dotg = ch[0] * gn[0] + ch[1] * gn[1] + ch[2] * gn[2];
                    double cb[3], d0b[3], d1b[3], neg[3];
                    for (int a = 0; a < 3; ++a) cb[a] = sgn * (gn[a] - ch[a] * dotg) / len;
                    cross(d1, cb, d0b);   /* c = d0 x d1: d0b = d1 x cb, d1b = cb x d0 */
                    cross(cb, d0, d1b);
                    for (int a = 0; a < 3; ++a) neg[a] = -(d0b[a] + d1b[a]);
                    add3(grad_pos, vi[1], d0b, 1.0); add3(grad_pos, vi[2], d1b, 1.0); add3(grad_pos, vi[0], neg, 1.0);
                }
                if (s->aux && grad_alpha) {
                    uint32_t bid = s->aux[4 * i];
                    if (bid < (uint64_t) B) {
                        const double *gm = out_param + ((int64_t) (5 * it + 4) * N + i) * 3;
                        const float *dh = (const float *) (s->aux + 4 * i + 1);
                        add1(grad_alpha + bid, gm[0] * dh[0] + gm[1] * dh[1] + gm[2] * dh[2]);
                    }
                }
            }
            /* epsm.py:609-620: occluder of the first vertex's emitter sample receives diffuse_grad[0] * dis */
            if (it == 0 && s->shadow) {
                uint32_t h[4];
                table_row(tri_table, T, s->shadow[4 * i], h);
                const float *hf = (const float *) (s->shadow + 4 * i + 1);
                if (h[0] < (uint64_t) V && h[1] < (uint64_t) V && h[2] < (uint64_t) V && (h[3] & EPSM_MODE_POS_ATTACHED)) {
                    double c0 = hf[0], c1 = hf[1], dis = hf[2];
                    const double *g = out_diffuse + ((int64_t) 0 * N + i) * 3;
                    add3(grad_pos, h[0], g, dis * c0); add3(grad_pos, h[1], g, dis * c1); add3(grad_pos, h[2], g, dis * (1.0 - c0 - c1));
                }
            }
            /* epsm.py:622-627 */
            if (s->emit) {
                uint32_t e[4];
                table_row(tri_table, T, s->emit[4 * i], e);
                const float *ef = (const float *) (s->emit + 4 * i + 1);
                /* si_direct.p is AD-attached only when the emitter mesh's positions are (dr.backward reaches nothing else) */
                if (e[0] < (uint64_t) V && e[1] < (uint64_t) V && e[2] < (uint64_t) V && (e[3] & EPSM_MODE_POS_ATTACHED)) {
                    double c0 = ef[0], c1 = ef[1], w = ef[2];
                    const double *g = out_light + ((int64_t) it * N + i) * 3;
                    add3(grad_pos, e[0], g, w * c0); add3(grad_pos, e[1], g, w * c1); add3(grad_pos, e[2], g, w * (1.0 - c0 - c1));
                }
            }
        }
    }
    }   /* omp parallel */
    if (priv) {
        g_private = 0;
#pragma omp parallel for schedule(static)
        for (int64_t j = 0; j < per_thread; ++j) {
            double s = 0.0;
            for (int t = 0; t < nthreads; ++t) s += priv[(int64_t) t * per_thread + j];
            if (j < 3 * V) out_pos[j] += s;
            else if (j < 6 * V) out_nrm[j - 3 * V] += s;
            else out_alpha[j - 6 * V] += s;
        }
        free(priv);
    }
    return 0;
}
