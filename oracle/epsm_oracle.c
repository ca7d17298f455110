/*
 * epsm_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * CPU restatement of the reference's `calc_grad` (both variants), written to
 * follow the reference line by line so that it can act as the parity oracle
 * for the HIP kernels.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library.  The product path (epsm_mitsuba3_amd)
 * never calls it and fails loudly if the HIP library is missing.
 *
 * Reference followed:  /root/reference/src/python/python/ad/integrators/epsm.py
 *     ManifoldIntegrator.calc_grad          :745-946
 *     ManifoldCausticIntegrator.calc_grad   :952-1200
 * The reference differentiates the constraint with torch autograd
 * (epsm.py:822-842, 885-907, 1029-1070, 1118-1163); this file restates the same
 * operation sequence and its reverse-mode derivative by hand (`halfvec_rev`,
 * `wo2_rev` below), builds the SAME dense `constraint` matrix and per-parameter
 * Jacobian rows, inverts `cur` with a dense LU with partial pivoting (what
 * torch.linalg.inv does through LAPACK getrf/getri, epsm.py:848,912,1076,1168),
 * and applies the same masks, nan_to_num and outlier clamp.
 *
 * Parity pinning: tests/test_oracle_golden.py checks this file against golden
 * vectors produced by the reference's own calc_grad (imported under stubs by
 * tests/golden/gen_golden.py), in fp32 and -- with the reference run under
 * torch.float64 -- in fp64 to ~1e-9, which pins every masking / overwrite
 * quirk of the reference independently of rounding.
 *
 * Build twice:  -DREAL=float -DSFX=f32   and   -DREAL=double -DSFX=f64.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/epsm.h"

#ifndef REAL
#define REAL float
#define SFX f32
#endif
#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SFX)

#define ORACLE_MAX_K 8
#define MAXL (ORACLE_MAX_K + 1)
#define MAXM (2 * MAXL)
#define MAXP (5 * ORACLE_MAX_K)

typedef REAL real;

static inline real real_max(void) { return sizeof(real) == 4 ? (real) FLT_MAX : (real) DBL_MAX; }

/* torch.nan_to_num (epsm.py:856 ...): nan -> 0, +-inf -> +-max */
static inline real nan_to_num(real x) {
    if (isnan(x)) return (real) 0;
    if (isinf(x)) return x > 0 ? real_max() : -real_max();
    return x;
}

static inline real dot3(const real *a, const real *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void cross3(const real *a, const real *b, real *c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
static inline real norm3(const real *a) { return (real) sqrt((double) dot3(a, a)) ; }

/* Frame of epsm.py:746-756 (create_local_frame): rows t, n^ x t, n^ with
 * t = normalize(0, -n^_z, n^_y). */
typedef struct {
    real nn[3], t[3], bt[3];
    real inv_n, inv_v;   /* 1/|n|, 1/|v| */
} frame_t;

static void make_frame(const real *n, frame_t *f) {
    real ln = norm3(n);
    f->inv_n = (real) 1 / ln;
    for (int c = 0; c < 3; ++c) f->nn[c] = n[c] / ln;
    real v[3] = {(real) 0, -f->nn[2], f->nn[1]};
    real lv = norm3(v);
    f->inv_v = (real) 1 / lv;
    for (int c = 0; c < 3; ++c) f->t[c] = v[c] / lv;
    cross3(f->nn, f->t, f->bt);
}

/* Reverse sweep through the frame: given adjoints of the three rows
 * (tb = d/dt, bb = d/dbt, nb = d/dnn) accumulate d/dn (un-normalised normal). */
static void frame_rev(const frame_t *f, const real *tb_in, const real *bb, const real *nb_in, real *gn) {
    real tb[3] = {tb_in[0], tb_in[1], tb_in[2]};
    real nb[3] = {nb_in[0], nb_in[1], nb_in[2]};
    real tmp[3];
    /* bt = cross(nn, t) */
    cross3(f->t, bb, tmp);
    for (int c = 0; c < 3; ++c) nb[c] += tmp[c];
    cross3(bb, f->nn, tmp);
    for (int c = 0; c < 3; ++c) tb[c] += tmp[c];
    /* t = v/|v| */
    real tt = dot3(f->t, tb);
    real vb[3];
    for (int c = 0; c < 3; ++c) vb[c] = (tb[c] - f->t[c] * tt) * f->inv_v;
    /* v = (0, -nn_z, nn_y) */
    nb[2] += -vb[1];
    nb[1] += vb[2];
    /* nn = n/|n| */
    real nnb = dot3(f->nn, nb);
    for (int c = 0; c < 3; ++c) gn[c] = (nb[c] - f->nn[c] * nnb) * f->inv_n;
}

typedef struct {
    real wi[3], wo[3];
    real inv_a, inv_b;      /* 1/|xp-xc|, 1/|xn-xc| */
    real res[3];            /* normalised half vector in the local frame */
    real inv_r;             /* 1/|wi2 + eta wo2| */
    real eta;
} halfvec_t;

/* Primal of epsm.py:809-821 / 875-883. */
static void halfvec_fwd(const real *xp, const real *xc, const real *xn, const frame_t *f, real eta, halfvec_t *h) {
    real a[3], b[3];
    for (int c = 0; c < 3; ++c) { a[c] = xp[c] - xc[c]; b[c] = xn[c] - xc[c]; }
    real la = norm3(a), lb = norm3(b);
    h->inv_a = (real) 1 / la;
    h->inv_b = (real) 1 / lb;
    for (int c = 0; c < 3; ++c) { h->wi[c] = a[c] / la; h->wo[c] = b[c] / lb; }
    real wi2[3] = {dot3(f->t, h->wi), dot3(f->bt, h->wi), dot3(f->nn, h->wi)};
    real wo2[3] = {dot3(f->t, h->wo), dot3(f->bt, h->wo), dot3(f->nn, h->wo)};
    real r[3];
    for (int c = 0; c < 3; ++c) r[c] = wi2[c] + wo2[c] * eta;
    real lr = norm3(r);
    h->inv_r = (real) 1 / lr;
    for (int c = 0; c < 3; ++c) h->res[c] = r[c] / lr;
    h->eta = eta;
}

/* d res[i] / d (xp, xc, xn, n).  `attached` = 0 reproduces the detached frame
 * of the caustic NEE sub-path (epsm.py:1022): gn is then zero. */
static void halfvec_rev(const frame_t *f, const halfvec_t *h, int i, int attached,
                        real *gxp, real *gxc, real *gxn, real *gn) {
    real eb[3] = {0, 0, 0};
    eb[i] = 1;
    real re = h->res[i];                       /* res . e_i */
    real rb[3];
    for (int c = 0; c < 3; ++c) rb[c] = (eb[c] - h->res[c] * re) * h->inv_r;
    real wi2b[3] = {rb[0], rb[1], rb[2]};
    real wo2b[3] = {rb[0] * h->eta, rb[1] * h->eta, rb[2] * h->eta};
    /* wi2 = R wi, wo2 = R wo  ->  wib = R^T wi2b ... */
    real wib[3], wob[3];
    for (int c = 0; c < 3; ++c) {
        wib[c] = f->t[c] * wi2b[0] + f->bt[c] * wi2b[1] + f->nn[c] * wi2b[2];
        wob[c] = f->t[c] * wo2b[0] + f->bt[c] * wo2b[1] + f->nn[c] * wo2b[2];
    }
    real wiw = dot3(h->wi, wib), wow = dot3(h->wo, wob);
    for (int c = 0; c < 3; ++c) {
        real ab = (wib[c] - h->wi[c] * wiw) * h->inv_a;
        real bb = (wob[c] - h->wo[c] * wow) * h->inv_b;
        gxp[c] = ab;
        gxn[c] = bb;
        gxc[c] = -ab - bb;
    }
    if (attached) {
        real tb[3], btb[3], nb[3];
        for (int c = 0; c < 3; ++c) {
            tb[c] = wi2b[0] * h->wi[c] + wo2b[0] * h->wo[c];
            btb[c] = wi2b[1] * h->wi[c] + wo2b[1] * h->wo[c];
            nb[c] = wi2b[2] * h->wi[c] + wo2b[2] * h->wo[c];
        }
        frame_rev(f, tb, btb, nb, gn);
    } else {
        gn[0] = gn[1] = gn[2] = 0;
    }
}

/* d wo2[i] / d (xc, xn, n) with wo2 = R normalize(xn - xc)  -- the caustic
 * pseudo-constraint res2 = wo2 - wo2.detach() (epsm.py:1028,1116). */
static void wo2_rev(const frame_t *f, const halfvec_t *h, int i, int attached,
                    real *gxc, real *gxn, real *gn) {
    const real *row = (i == 0) ? f->t : (i == 1 ? f->bt : f->nn);
    real wob[3] = {row[0], row[1], row[2]};
    real wow = dot3(h->wo, wob);
    for (int c = 0; c < 3; ++c) {
        real bb = (wob[c] - h->wo[c] * wow) * h->inv_b;
        gxn[c] = bb;
        gxc[c] = -bb;
    }
    if (attached) {
        real z[3] = {0, 0, 0};
        const real *tb = (i == 0) ? h->wo : z;
        const real *btb = (i == 1) ? h->wo : z;
        const real *nb = (i == 2) ? h->wo : z;
        frame_rev(f, tb, btb, nb, gn);
    } else {
        gn[0] = gn[1] = gn[2] = 0;
    }
}

/* Dense inverse by LU with partial pivoting (LAPACK getrf + column solves);
 * a is n x n row-major with leading dimension MAXM; result in inv (same ld).
 * A zero pivot is divided by like any other (torch raises instead); the
 * resulting inf/nan are handled by the caller's nan_to_num. */
static void dense_inverse(int n, real a[MAXM][MAXM], real inv[MAXM][MAXM]) {
    int piv[MAXM];
    real lu[MAXM][MAXM];
    for (int r = 0; r < n; ++r) for (int c = 0; c < n; ++c) lu[r][c] = a[r][c];
    for (int r = 0; r < n; ++r) piv[r] = r;
    for (int k = 0; k < n; ++k) {
        int p = k;
        real best = (real) fabs((double) lu[k][k]);
        for (int r = k + 1; r < n; ++r) {
            real v = (real) fabs((double) lu[r][k]);
            if (v > best) { best = v; p = r; }
        }
        if (p != k) {
            for (int c = 0; c < n; ++c) { real t = lu[k][c]; lu[k][c] = lu[p][c]; lu[p][c] = t; }
            int t = piv[k]; piv[k] = piv[p]; piv[p] = t;
        }
        real d = lu[k][k];
        for (int r = k + 1; r < n; ++r) {
            real m = lu[r][k] / d;
            lu[r][k] = m;
            for (int c = k + 1; c < n; ++c) lu[r][c] -= m * lu[k][c];
        }
    }
    for (int col = 0; col < n; ++col) {
        real y[MAXM];
        for (int r = 0; r < n; ++r) {        /* L y = P e_col */
            real s = (piv[r] == col) ? (real) 1 : (real) 0;
            for (int c = 0; c < r; ++c) s -= lu[r][c] * y[c];
            y[r] = s;
        }
        for (int r = n - 1; r >= 0; --r) {   /* U x = y */
            real s = y[r];
            for (int c = r + 1; c < n; ++c) s -= lu[r][c] * inv[c][col];
            inv[r][col] = s / lu[r][r];
        }
    }
}

/* 2-norm condition number of the system a solve used, from the matrix and its computed inverse: largest singular
 * values by power iteration on M^T M (n <= 10; an under-estimate if anything).  Test infrastructure of the test
 * infrastructure: SURVEY.md 8c states the fp32 tolerance for paths with cond_2 < 1e4, so the parity tests need to know
 * which paths those are.  Non-finite entries -> +inf. */
static double norm2_est(int n, real m[MAXM][MAXM]) {
    double x[MAXM], y[MAXM], z[MAXM], s = 0.0;
    for (int r = 0; r < n; ++r) { x[r] = 1.0 + 0.37 * r; for (int c = 0; c < n; ++c) if (!isfinite((double) m[r][c])) return INFINITY; }
    for (int it = 0; it < 80; ++it) {
        for (int r = 0; r < n; ++r) { double a = 0; for (int c = 0; c < n; ++c) a += (double) m[r][c] * x[c]; y[r] = a; }
        for (int c = 0; c < n; ++c) { double a = 0; for (int r = 0; r < n; ++r) a += (double) m[r][c] * y[r]; z[c] = a; }
        double l = 0; for (int c = 0; c < n; ++c) l += z[c] * z[c];
        l = sqrt(l);
        if (!(l > 0)) return 0.0;
        for (int c = 0; c < n; ++c) x[c] = z[c] / l;
        s = sqrt(l);
    }
    return s;
}
static double cond2_est(int n, real a[MAXM][MAXM], real inv[MAXM][MAXM]) {
    const double c = norm2_est(n, a) * norm2_est(n, inv);
    return c == c ? c : INFINITY;
}
static double *FN(g_cond_out) = 0;
/* buf: N doubles that the next calc_grad calls fill with max cond_2 over the solves whose results a path USES
 * (1 for paths that use none); NULL switches it off again. */
void FN(epsm_oracle_set_cond_out)(double *buf) { FN(g_cond_out) = buf; }

typedef struct {
    const real *p[3], *n[3], *light;
    real b0, b1, eta;
    uint32_t bsdf;
    int active, active_em, ismesh;
} vtx_t;

static void load_vertex(const EpsmVertexRecord *v, int64_t i, vtx_t *o) {
    o->p[0] = (const real *) v->p0 + 3 * i;
    o->p[1] = (const real *) v->p1 + 3 * i;
    o->p[2] = (const real *) v->p2 + 3 * i;
    o->n[0] = (const real *) v->n0 + 3 * i;
    o->n[1] = (const real *) v->n1 + 3 * i;
    o->n[2] = (const real *) v->n2 + 3 * i;
    o->light = (const real *) v->light + 3 * i;
    o->b0 = ((const real *) v->b0)[i];
    o->b1 = ((const real *) v->b1)[i];
    o->eta = ((const real *) v->eta)[i];
    o->bsdf = v->bsdf[i];
    o->active = v->active[i] != 0;
    o->active_em = v->active_em[i] != 0;
    o->ismesh = v->ismesh[i] != 0;
}

/* epsm.py:758-762 get_point / get_normal */
static void interp3(const real *const a[3], real b0, real b1, real *o) {
    real b2 = (real) 1 - b0 - b1;
    for (int c = 0; c < 3; ++c) o[c] = a[0][c] * b0 + a[1][c] * b1 + a[2][c] * b2;
}

typedef struct {
    real constraint[MAXM][MAXM];     /* epsm.py:771 */
    real pg[MAXP][MAXM][3];          /* param_grad_list, epsm.py:768 */
    real fin[MAXP][3];               /* final_param_grad */
} work_t;

/* Scatter the position gradient `g` of an interpolated point into the rows of
 * its three triangle-vertex parameters (d point / d p_j = b_j I). */
static void set_point_param_rows(work_t *w, int pidx0, int row, const vtx_t *v, const real *g) {
    real bw[3] = {v->b0, v->b1, (real) 1 - v->b0 - v->b1};
    for (int j = 0; j < 3; ++j)
        for (int c = 0; c < 3; ++c) w->pg[pidx0 + j][row][c] = bw[j] * g[c];
}

static void uv_grad(const real *const a[3], const real *g, real *o /*2*/) {
    real e0[3], e1[3];
    for (int c = 0; c < 3; ++c) { e0[c] = a[0][c] - a[2][c]; e1[c] = a[1][c] - a[2][c]; }
    o[0] = dot3(g, e0);
    o[1] = dot3(g, e1);
}

static void one_path(int variant, int K, int64_t i, const real *cam_all, const EpsmVertexRecord *verts,
                     const real *dlduv_row, int dlduv_cols, const real *dldp_in, real clip,
                     real *out_param, real *out_light, real *out_diffuse, int64_t N) {
    const int caustic = variant == EPSM_VARIANT_MANIFOLD_CAUSTIC;
    const int L = K + 1, M = 2 * L;
    work_t w;
    memset(&w, 0, sizeof(w));
    double cond = 1.0;
    vtx_t v[ORACLE_MAX_K + 2];
    for (int k = 1; k <= K; ++k) load_vertex(&verts[k - 1], i, &v[k]);
    const real *cam = cam_all + 3 * i;

    real dlduv[MAXM];
    for (int c = 0; c < M; ++c) dlduv[c] = (c < dlduv_cols) ? dlduv_row[c] : (real) 0;
    real dldp[3] = {dldp_in[0], dldp_in[1], dldp_in[2]};

    real light_grad[ORACLE_MAX_K][3], diffuse_grad[ORACLE_MAX_K][3];
    memset(light_grad, 0, sizeof(light_grad));
    memset(diffuse_grad, 0, sizeof(diffuse_grad));

    /* index of the parameters of vertex k in param_list: p0,p1,p2 at 5(k-1),
     * n at 5(k-1)+3, m at 5(k-1)+4 (epsm.py:786-788,815-816 / 993-995,1104-1105). */
    int nparam = 0;          /* current len(param_list) */
    int valid = 0;
    real hasdiffuse = 0;
    int diffuse_pos = 0;
    real point_prev[3], point_cur[3], point_next[3], nrm[3];

    for (int id = 1; id <= K; ++id) {
        const int isdiffuse = (v[id].bsdf & EPSM_BSDF_DIFFUSE) != 0;     /* :781 */
        const int P0 = 5 * (id - 1);
        nparam = P0 + 3;                                                  /* :786-788 */
        if (id == 1) {
            for (int c = 0; c < 3; ++c) point_prev[c] = cam[c];
            if (!isdiffuse) { dldp[0] = dldp[1] = dldp[2] = 0; }         /* :791 */
            if (caustic && !isdiffuse) for (int c = 0; c < M; ++c) dlduv[c] = 0;  /* :999 */
            for (int c = 0; c < 3; ++c) diffuse_grad[0][c] = dldp[c];    /* :792 */
            valid = v[id].ismesh;                                         /* :793 */
        } else {
            interp3(v[id - 1].p, v[id - 1].b0, v[id - 1].b1, point_prev); /* :795 */
            valid = valid && v[id].ismesh;                                /* :796 */
        }
        hasdiffuse += isdiffuse ? 1 : 0;                                  /* :799 */
        valid = valid && (hasdiffuse < 2);                                /* :800 */
        if (isdiffuse) diffuse_pos = id;                                  /* :801 */
        const int nolight = !v[id].active_em;                             /* :802 */
        interp3(v[id].p, v[id].b0, v[id].b1, point_cur);                  /* :803 */
        interp3(v[id].n, v[id].b0, v[id].b1, nrm);                        /* :813 / :1021 */
        frame_t fr;
        make_frame(nrm, &fr);
        const int r0 = 2 * id - 2;

        /* ---------------- light-sampling sub-path (A) ---------------- */
        real plg[MAXM][3];                                                /* param_light_grad :808 */
        memset(plg, 0, sizeof(plg));
        if (!caustic) nparam = P0 + 5;                                    /* add(n); add(m) :815-816 */
        {
            halfvec_t h;
            halfvec_fwd(point_prev, point_cur, v[id].light, &fr, v[id].eta, &h);
            const int attached = !caustic;                                /* :1022 detach */
            for (int ii = 0; ii < 2; ++ii) {
                real gxp[3], gxc[3], gxn[3], gn[3], g2[2];
                halfvec_rev(&fr, &h, ii, attached, gxp, gxc, gxn, gn);
                const int row = r0 + ii;
                if (id > 1) {                                             /* :825-829 */
                    uv_grad(v[id - 1].p, gxp, g2);
                    w.constraint[row][2 * id - 2] = g2[0];
                    w.constraint[row][2 * id - 1] = g2[1];
                }
                uv_grad(v[id].p, gxc, g2);
                if (attached) { real gnuv[2]; uv_grad(v[id].n, gn, gnuv); g2[0] += gnuv[0]; g2[1] += gnuv[1]; }
                w.constraint[row][2 * id + 0] = g2[0];                    /* :830-831 */
                w.constraint[row][2 * id + 1] = g2[1];
                /* :835-840  every registered parameter: grad if in graph else 0 */
                for (int q = 0; q < nparam; ++q) for (int c = 0; c < 3; ++c) w.pg[q][row][c] = 0;
                if (id > 1) set_point_param_rows(&w, P0 - 5, row, &v[id - 1], gxp);
                set_point_param_rows(&w, P0, row, &v[id], gxc);
                if (!caustic) for (int c = 0; c < 3; ++c) w.pg[P0 + 3][row][c] = gn[c];
                for (int c = 0; c < 3; ++c) plg[row][c] = gxn[c];         /* :841 */

                if (caustic) {                                            /* :1051-1070 */
                    real hxc[3], hxn[3], hn[3];
                    wo2_rev(&fr, &h, ii, 0, hxc, hxn, hn);
                    int light_grad_alive = 1;   /* point_next.grad is reset inside the j loop (:1064) */
                    for (int j = 1; j <= id; ++j) {
                        if (diffuse_pos == j) {
                            const int rj = 2 * j - 2 + ii;
                            for (int c = 0; c < M; ++c) w.constraint[rj][c] = 0;
                            uv_grad(v[id].p, hxc, g2);
                            w.constraint[rj][2 * id + 0] = g2[0];
                            w.constraint[rj][2 * id + 1] = g2[1];
                            for (int q = 0; q < nparam; ++q) for (int c = 0; c < 3; ++c) w.pg[q][rj][c] = 0;
                            set_point_param_rows(&w, P0, rj, &v[id], hxc);
                            for (int c = 0; c < 3; ++c) plg[rj][c] = light_grad_alive ? hxn[c] : (real) 0;
                        }
                        /* :1062-1066 -- the reset happens for every j once the grad exists */
                        light_grad_alive = 0;
                    }
                }
            }
        }
        {
            real cur[MAXM][MAXM], inv[MAXM][MAXM];
            const int n = 2 * id;
            const int ident = (!valid) || (!v[id].active) || nolight;     /* :845-847 */
            for (int r = 0; r < n; ++r) for (int c = 0; c < n; ++c)
                cur[r][c] = ident ? (real) (r == c) : w.constraint[r][c + 2];
            dense_inverse(n, cur, inv);                                   /* :848 */
            real y[MAXM];   /* dlduv[:2id] . inv */
            for (int c = 0; c < n; ++c) { real s = 0; for (int r = 0; r < n; ++r) s += dlduv[r] * inv[r][c]; y[c] = s; }
            const int masked = (!valid) || (!v[id].active) || nolight || (hasdiffuse > 0);  /* :852-855 */
            if (FN(g_cond_out) && !masked) { const double cc = cond2_est(n, cur, inv); if (!(cc <= cond)) cond = cc; }
            for (int q = 0; q < nparam; ++q)
                for (int c = 0; c < 3; ++c) {
                    real s = 0;
                    for (int r = 0; r < n; ++r) s += y[r] * (-w.pg[q][r][c]);
                    if (masked) s = 0;
                    w.fin[q][c] += nan_to_num(s);                         /* :856-857 */
                }
            for (int c = 0; c < 3; ++c) {                                 /* :859-866 */
                real s = 0;
                for (int r = 0; r < n; ++r) s += y[r] * (-plg[r][c]);
                if (masked) s = 0;
                light_grad[id - 1][c] = nan_to_num(s);
            }
        }

        /* ---------------- continuing sub-path (B) ---------------- */
        if (id < K) {
            interp3(v[id + 1].p, v[id + 1].b0, v[id + 1].b1, point_next); /* :873 */
            if (caustic) nparam = P0 + 5;                                 /* :1104-1105 */
            real pdg[MAXM][3];                                            /* param_diffuse_grad :884 */
            memset(pdg, 0, sizeof(pdg));
            halfvec_t h;
            halfvec_fwd(point_prev, point_cur, point_next, &fr, v[id].eta, &h);
            real leftover[3] = {0, 0, 0};   /* caustic: point_next.grad is not reset after res2 (:1154-1163) */
            for (int ii = 0; ii < 2; ++ii) {
                real gxp[3], gxc[3], gxn[3], gn[3], g2[2], gnuv[2];
                halfvec_rev(&fr, &h, ii, 1, gxp, gxc, gxn, gn);
                const int row = r0 + ii;
                if (id > 1) {                                             /* :888-892 */
                    uv_grad(v[id - 1].p, gxp, g2);
                    w.constraint[row][2 * id - 2] = g2[0];
                    w.constraint[row][2 * id - 1] = g2[1];
                }
                uv_grad(v[id].p, gxc, g2);
                uv_grad(v[id].n, gn, gnuv);
                w.constraint[row][2 * id + 0] = g2[0] + gnuv[0];          /* :893-894 */
                w.constraint[row][2 * id + 1] = g2[1] + gnuv[1];
                uv_grad(v[id + 1].p, gxn, g2);
                w.constraint[row][2 * id + 2] = g2[0];                    /* :895-896 */
                w.constraint[row][2 * id + 3] = g2[1];
                /* :901-904  only parameters that received a gradient are overwritten */
                if (id > 1) set_point_param_rows(&w, P0 - 5, row, &v[id - 1], gxp);
                set_point_param_rows(&w, P0, row, &v[id], gxc);
                for (int c = 0; c < 3; ++c) w.pg[P0 + 3][row][c] = gn[c];
                for (int c = 0; c < 3; ++c) w.pg[P0 + 4][row][c] = (c == ii) ? (real) -1 : (real) 0;   /* res = ... - m */
                for (int c = 0; c < 3; ++c) pdg[row][c] = gxn[c] + leftover[c];   /* :906 / :1139 */

                if (caustic) {                                            /* :1141-1163 */
                    real hxc[3], hxn[3], hn[3];
                    wo2_rev(&fr, &h, ii, 1, hxc, hxn, hn);
                    for (int j = 1; j <= id; ++j) {
                        if (diffuse_pos == j) {
                            const int rj = 2 * j - 2 + ii;
                            for (int c = 0; c < M; ++c) w.constraint[rj][c] = 0;
                            uv_grad(v[id].p, hxc, g2);
                            uv_grad(v[id].n, hn, gnuv);
                            w.constraint[rj][2 * id + 0] = g2[0] + gnuv[0];
                            w.constraint[rj][2 * id + 1] = g2[1] + gnuv[1];
                            uv_grad(v[id + 1].p, hxn, g2);
                            w.constraint[rj][2 * id + 2] = g2[0];
                            w.constraint[rj][2 * id + 3] = g2[1];
                            for (int q = 0; q < nparam; ++q) for (int c = 0; c < 3; ++c) w.pg[q][rj][c] = 0;
                            set_point_param_rows(&w, P0, rj, &v[id], hxc);
                            for (int c = 0; c < 3; ++c) w.pg[P0 + 3][rj][c] = hn[c];
                            for (int c = 0; c < 3; ++c) pdg[rj][c] = hxn[c];
                        }
                    }
                    /* point_next.grad keeps d wo2[ii] and the next backward accumulates onto it */
                    for (int c = 0; c < 3; ++c) leftover[c] = hxn[c];
                }
            }
            real cur[MAXM][MAXM], inv[MAXM][MAXM];
            const int n = 2 * id;
            const int ident = (!valid) || (!v[id + 1].active);            /* :910-911 */
            for (int r = 0; r < n; ++r) for (int c = 0; c < n; ++c)
                cur[r][c] = ident ? (real) (r == c) : w.constraint[r][c + 2];
            dense_inverse(n, cur, inv);                                   /* :912 */
            real y[MAXM];
            for (int c = 0; c < n; ++c) { real s = 0; for (int r = 0; r < n; ++r) s += dlduv[r] * inv[r][c]; y[c] = s; }
            const int next_diffuse = (v[id + 1].bsdf & EPSM_BSDF_DIFFUSE) != 0;
            const int next_null = (v[id + 1].bsdf & EPSM_BSDF_NULL) != 0;
            int masked_p, masked_d;
            if (!caustic) {                                               /* :916-920, :925-928 */
                masked_p = (!valid) || (!v[id + 1].active) || (!next_diffuse) || (hasdiffuse > 0);
                masked_d = masked_p;
            } else {                                                      /* :1172-1174, :1180-1182 */
                masked_p = (!valid) || (!v[id + 1].active) || (!next_diffuse);
                masked_d = (!valid) || (!v[id + 1].active) || ((!next_null) && (!next_diffuse));
            }
            if (FN(g_cond_out) && !(masked_p && masked_d)) { const double cc = cond2_est(n, cur, inv); if (!(cc <= cond)) cond = cc; }
            for (int q = 0; q < nparam; ++q)
                for (int c = 0; c < 3; ++c) {
                    real s = 0;
                    for (int r = 0; r < n; ++r) s += y[r] * (-w.pg[q][r][c]);
                    if (masked_p) s = 0;
                    w.fin[q][c] += nan_to_num(s);
                }
            for (int c = 0; c < 3; ++c) {
                real s = 0;
                for (int r = 0; r < n; ++r) s += y[r] * (-pdg[r][c]);
                if (masked_d) s = 0;
                diffuse_grad[id][c] = nan_to_num(s);
            }
        }
    }

    if (FN(g_cond_out)) FN(g_cond_out)[i] = cond;

    /* remove outlier (epsm.py:932-944 / 1186-1198) */
    const int do_clip = (clip > 0) && !isinf(clip);
    const int P = epsm_num_param_grads(variant, K);
    for (int q = 0; q < P; ++q)
        for (int c = 0; c < 3; ++c) {
            real g = w.fin[q][c];
            if (do_clip && (g > clip || g < -clip)) g = 0;
            out_param[((int64_t) q * N + i) * 3 + c] = g;
        }
    for (int k = 0; k < K; ++k)
        for (int c = 0; c < 3; ++c) {
            real g = light_grad[k][c];
            if (do_clip && (g > clip || g < -clip)) g = 0;
            out_light[((int64_t) k * N + i) * 3 + c] = g;
            g = diffuse_grad[k][c];
            if (do_clip && (g > clip || g < -clip)) g = 0;
            out_diffuse[((int64_t) k * N + i) * 3 + c] = g;
        }
}

#ifndef EPSM_ORACLE_NO_COMMON
int epsm_num_param_grads(int variant, int K) {
    return variant == EPSM_VARIANT_MANIFOLD_CAUSTIC ? 5 * K - 2 : 5 * K;
}
#endif

/* Same argument meaning as epsm_manifold_grad (include/epsm.h) with HOST
 * pointers of element type REAL; `nthreads` <= 0 uses every core OpenMP sees.
 * Returns the number of threads used, or a negative EPSM_E* code. */
int FN(epsm_oracle_calc_grad)(int variant, int64_t N, int K,
                              const void *cam, const EpsmVertexRecord *verts,
                              const void *dlduv, int64_t dlduv_stride, int dlduv_cols,
                              const void *dldp, double clip,
                              void *out_param, void *out_light, void *out_diffuse, int nthreads) {
    if (K < 1 || K > ORACLE_MAX_K || N < 0 || !cam || !verts || !dlduv || !dldp ||
        !out_param || !out_light || !out_diffuse)
        return EPSM_EINVAL;
    if (variant != EPSM_VARIANT_MANIFOLD && variant != EPSM_VARIANT_MANIFOLD_CAUSTIC) return EPSM_EINVAL;
    int used = 1;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
    used = nthreads;
#pragma omp parallel for schedule(static, 256) num_threads(nthreads)
#endif
    for (int64_t i = 0; i < N; ++i)
        one_path(variant, K, i, (const real *) cam, verts,
                 (const real *) dlduv + i * dlduv_stride, dlduv_cols,
                 (const real *) dldp + 3 * i, (real) clip,
                 (real *) out_param, (real *) out_light, (real *) out_diffuse, N);
    return used;
}
