// epsm_tangent_core.h -- per-path first-vertex tangent (epsm.py:250-272), shared by the stand-alone kernel
// (epsm_tangent.hip) and the backward-pass kernel that computes it in place (epsm_grad_scatter.hip).
//
// Closed form of the forward-mode AD block of render_backward: the directional derivative of the
// Moeller-Trumbore barycentrics (include/mitsuba/render/mesh.h:343-365) along the image-space motion
//     grad_d = (d_x - d) gx + (d_y - d) gy                                   (epsm.py:255)
// mapped to (b0, b1) and the hit point as src/render/mesh.cpp:698-709 does.
#pragma once

#include "epsm_path_core.h"

namespace epsm {

struct TangentIn {
    int64_t path_offset;
    int spp, res, img_width, img_channels;
    const float *o, *d, *dx, *dy, *grad_img;
};
struct Tangent { float db0, db1; V3<float> dp, gd; };     // d b0, d b1, d si.p, grad_d

// the arithmetic, on values; `gd` = the image-space motion of the ray direction, (d_x - d) gx + (d_y - d) gy  (epsm.py:255)
EPSM_HD Tangent tangent_from_gd(V3<float> o, V3<float> d, V3<float> gd, V3<float> p0, V3<float> p1, V3<float> p2, bool active) {
    Tangent t;
    t.gd = gd;
    t.db0 = t.db1 = 0.f;
    t.dp = zero3<float>();
    if (active) {
        const V3<float> e1 = p1 - p0, e2 = p2 - p0;               // mesh.h:349
        const V3<float> pvec = cross(d, e2);
        const float inv_det = rcp_(dot(e1, pvec));
        const V3<float> tvec = o - p0;
        const float u = dot(tvec, pvec) * inv_det;
        const V3<float> qvec = cross(tvec, e1);
        const float v = dot(d, qvec) * inv_det;
        // forward derivative along gd (ray origin fixed)
        const V3<float> dpvec = cross(t.gd, e2);
        const float ddet = dot(e1, dpvec);
        const float du = (dot(tvec, dpvec) - u * ddet) * inv_det;
        const float dv = (dot(t.gd, qvec) - v * ddet) * inv_det;
        t.db1 = du;                                                // b1 = prim_uv.x  (mesh.cpp:698)
        t.db0 = -du - dv;                                          // b0 = 1 - b1 - b2
        t.dp = e1 * du + e2 * dv;                                  // d (p0 b0 + p1 b1 + p2 b2)
    }
    return t;
}
EPSM_HD Tangent tangent_from(V3<float> o, V3<float> d, V3<float> dx, V3<float> dy, float gx, float gy,
                             V3<float> p0, V3<float> p1, V3<float> p2, bool active) {
    return tangent_from_gd(o, d, (dx - d) * gx + (dy - d) * gy, p0, p1, p2, active);
}

EPSM_HD Tangent first_vertex_tangent(const TangentIn &A, int64_t i, const float *p0a, const float *p1a, const float *p2a,
                                     bool active) {
    const int64_t pix = (A.path_offset + i) / A.spp;
    const int64_t y = pix / A.res, x = pix % A.res;
    const auto *g = gl(A.grad_img) + (y * A.img_width + x) * A.img_channels;
    const float gx = g[3], gy = g[4];
    const V3<float> d = load3(A.d, i), dx = load3(A.dx, i), dy = load3(A.dy, i);
    V3<float> o = zero3<float>(), p0 = o, p1 = o, p2 = o;
    if (active) { o = load3(A.o, i); p0 = load3(p0a, i); p1 = load3(p1a, i); p2 = load3(p2a, i); }
    return tangent_from(o, d, dx, dy, gx, gy, p0, p1, p2, active);
}

}  // namespace epsm
