/*
 * epsm_oracle_trace.c -- TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * A brute-force float64 ray / triangle-soup intersector: every ray against EVERY triangle, no acceleration
 * structure, the Moeller-Trumbore test of include/mitsuba/render/mesh.h:343-365 restated in double.  It is the
 * non-self oracle of the native tracer: the tracer's BVH build, traversal, leaf
 * tests and vertex log are checked against it by REPLAY -- the logged rays and vertices of a trace are re-intersected
 * here and must name the same primitive (tests/test_tracer_oracle.py, tests/test_gpu_tracer_oracle.py):
 *   closest hit   camera ray -> first vertex, vertex k -> vertex k+1 (origin = logged point, direction = towards the
 *                 next logged point): primitive index and (t, b0, b1);
 *   any hit       vertex k -> its emitter sample point: visibility, which the log carries as a zeroed emitter weight.
 *
 * PARITY UNPINNED by the reference (Mitsuba's scene.ray_intersect needs Embree / OptiX, neither can be built here);
 * pinned by construction: it is the definition of "closest hit" over a triangle soup.
 */
#include <math.h>
#include <stdint.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* mesh.h:343-365 in float64.  Returns 1 on a hit with t in [tmin, tmax]. */
static int moeller_trumbore(const double *o, const double *d, const double *q, double tmin, double tmax,
                            double *t_out, double *u_out, double *v_out) {
    const double *p0 = q, *p1 = q + 3, *p2 = q + 6;
    double e1[3], e2[3], pvec[3], tvec[3], qvec[3];
    for (int c = 0; c < 3; ++c) { e1[c] = p1[c] - p0[c]; e2[c] = p2[c] - p0[c]; tvec[c] = o[c] - p0[c]; }
    pvec[0] = d[1] * e2[2] - d[2] * e2[1]; pvec[1] = d[2] * e2[0] - d[0] * e2[2]; pvec[2] = d[0] * e2[1] - d[1] * e2[0];
    const double det = e1[0] * pvec[0] + e1[1] * pvec[1] + e1[2] * pvec[2];
    if (det == 0.0) return 0;
    const double inv_det = 1.0 / det;
    const double u = (tvec[0] * pvec[0] + tvec[1] * pvec[1] + tvec[2] * pvec[2]) * inv_det;
    if (!(u >= 0.0 && u <= 1.0)) return 0;
    qvec[0] = tvec[1] * e1[2] - tvec[2] * e1[1]; qvec[1] = tvec[2] * e1[0] - tvec[0] * e1[2]; qvec[2] = tvec[0] * e1[1] - tvec[1] * e1[0];
    const double v = (d[0] * qvec[0] + d[1] * qvec[1] + d[2] * qvec[2]) * inv_det;
    if (!(v >= 0.0 && u + v <= 1.0)) return 0;
    const double t = (e2[0] * qvec[0] + e2[1] * qvec[1] + e2[2] * qvec[2]) * inv_det;
    if (!(t >= tmin && t <= tmax)) return 0;
    *t_out = t; *u_out = u; *v_out = v;
    return 1;
}

/* n rays (o, d: (n,3); tmin, tmax: (n)) against T triangles (verts: (T,9) = p0,p1,p2).  skip[i] (may be NULL): a
 * triangle the ray does not see (the one it starts on), -1 = none.  any_hit != 0: stop at the first hit found (the
 * answer is hit / no hit; hit_tri is then SOME occluder).  Outputs: hit_tri (n) = index or -1, hit_t / hit_u / hit_v
 * (n; u, v are mesh.h's, i.e. the barycentric weights of p1 and p2).  Also: second_t (n, may be NULL) = distance of the
 * nearest hit on ANOTHER triangle than hit_tri (+inf if none) -- how close the runner-up was, for tie analysis. */
int epsm_oracle_intersect(int64_t n, const double *o, const double *d, const double *tmin, const double *tmax,
                          int64_t T, const double *verts, const int64_t *skip, int any_hit,
                          int64_t *hit_tri, double *hit_t, double *hit_u, double *hit_v, double *second_t) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < n; ++i) {
        int64_t best = -1;
        double bt = INFINITY, bu = 0, bv = 0, st = INFINITY;
        const int64_t sk = skip ? skip[i] : -1;
        for (int64_t k = 0; k < T; ++k) {
            if (k == sk) continue;
            double t, u, v;
            if (!moeller_trumbore(o + 3 * i, d + 3 * i, verts + 9 * k, tmin[i], tmax[i], &t, &u, &v)) continue;
            if (t < bt) { st = bt; best = k; bt = t; bu = u; bv = v; if (any_hit) break; }
            else if (t < st) st = t;
        }
        hit_tri[i] = best; hit_t[i] = bt; hit_u[i] = bu; hit_v[i] = bv;
        if (second_t) second_t[i] = st;
    }
    return 0;
}
