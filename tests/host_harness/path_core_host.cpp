// TEST HARNESS ONLY: compiles epsm_mitsuba3_amd/csrc/epsm_path_core.h -- the exact
// per-path code the gfx950 kernels run -- for the host CPU, in fp32 and fp64, so
// that the block-adjoint algebra can be checked against the oracle and the golden
// vectors on machines without a GPU (`pytest -m "not gpu"`).  Not shipped, not a
// fallback: the product library (libepsm_hip.so) exports the HIP path only.
#include "../../epsm_mitsuba3_amd/csrc/epsm_path_core.h"
#include "../../epsm_mitsuba3_amd/csrc/epsm_tangent_core.h"
#include "../../include/epsm.h"

using namespace epsm;

template <typename R>
static GradArgs<R> make_args(int64_t N, int K, const void *cam, const EpsmVertexRecord *verts,
                             const void *dlduv, int64_t stride, const void *dldp, double clip,
                             void *op, void *ol, void *od) {
    GradArgs<R> A{};
    A.N = N;
    A.cam = (const R *) cam;
    for (int k = 0; k < K; ++k) {
        const EpsmVertexRecord &v = verts[k];
        VertexPtrs<R> &o = A.v[k];
        o.p0 = (const R *) v.p0; o.p1 = (const R *) v.p1; o.p2 = (const R *) v.p2;
        o.n0 = (const R *) v.n0; o.n1 = (const R *) v.n1; o.n2 = (const R *) v.n2;
        o.b0 = (const R *) v.b0; o.b1 = (const R *) v.b1; o.eta = (const R *) v.eta;
        o.light = (const R *) v.light;
        o.bsdf = v.bsdf; o.active = v.active; o.active_em = v.active_em; o.ismesh = v.ismesh;
    }
    A.dlduv = (const R *) dlduv;
    A.dlduv_stride = stride;
    A.dldp = (const R *) dldp;
    A.clip = (clip > 0 && clip < 1e300) ? (R) clip : realmax_(R(0));
    A.out_param = (R *) op; A.out_light = (R *) ol; A.out_diffuse = (R *) od;
    return A;
}

template <typename R, int K>
static void run_k(int variant, bool full_d, const GradArgs<R> &A, int dcols) {
    for (int64_t i = 0; i < A.N; ++i) {
        const DenseOut<R> out{A, i};
        if (variant == EPSM_VARIANT_MANIFOLD) {
            if (full_d) manifold_path<R, K, true>(A, i, dcols, out); else manifold_path<R, K, false>(A, i, dcols, out);
        } else {
            if (full_d) caustic_path<R, K, true>(A, i, dcols, out); else caustic_path<R, K, false>(A, i, dcols, out);
        }
    }
}

template <typename R>
static int run(int variant, int64_t N, int K, const void *cam, const EpsmVertexRecord *verts,
               const void *dlduv, int64_t stride, int dcols, const void *dldp, double clip,
               void *op, void *ol, void *od) {
    if (K < 1 || K > kMaxVertices) return EPSM_EINVAL;
    GradArgs<R> A = make_args<R>(N, K, cam, verts, dlduv, stride, dldp, clip, op, ol, od);
    if (dcols > 2 * K) dcols = 2 * K;
    bool full_d = dcols > 2;
    switch (K) {
        case 1: run_k<R, 1>(variant, full_d, A, dcols); break;
        case 2: run_k<R, 2>(variant, full_d, A, dcols); break;
        case 3: run_k<R, 3>(variant, full_d, A, dcols); break;
        case 4: run_k<R, 4>(variant, full_d, A, dcols); break;
        case 5: run_k<R, 5>(variant, full_d, A, dcols); break;
    }
    return 0;
}

extern "C" int epsm_host_core_grad_f32(int variant, int64_t N, int K, const void *cam, const EpsmVertexRecord *verts,
                                       const void *dlduv, int64_t stride, int dcols, const void *dldp, double clip,
                                       void *op, void *ol, void *od, int) {
    return run<float>(variant, N, K, cam, verts, dlduv, stride, dcols, dldp, clip, op, ol, od);
}
extern "C" int epsm_host_core_grad_f64(int variant, int64_t N, int K, const void *cam, const EpsmVertexRecord *verts,
                                       const void *dlduv, int64_t stride, int dcols, const void *dldp, double clip,
                                       void *op, void *ol, void *od, int) {
    return run<double>(variant, N, K, cam, verts, dlduv, stride, dcols, dldp, clip, op, ol, od);
}

// the first-vertex tangent's arithmetic (epsm_tangent_core.h, what epsm_tangent_kernel and the backward kernel run) on one
// path: in = o, d, d_x, d_y (3 each), gx, gy, p0, p1, p2 (3 each); out = d b0, d b1, d p (3), grad_d (3)
extern "C" int epsm_host_tangent_from(const float *in, float *out) {
    auto v = [&](int j) { V3<float> r; r.x = in[j]; r.y = in[j + 1]; r.z = in[j + 2]; return r; };
    const Tangent t = tangent_from(v(0), v(3), v(6), v(9), in[12], in[13], v(14), v(17), v(20), true);
    out[0] = t.db0; out[1] = t.db1; out[2] = t.dp.x; out[3] = t.dp.y; out[4] = t.dp.z; out[5] = t.gd.x; out[6] = t.gd.y; out[7] = t.gd.z;
    return 0;
}
