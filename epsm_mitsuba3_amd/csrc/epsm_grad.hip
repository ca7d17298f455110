// epsm_grad.hip -- gfx950 kernel + C ABI entry of calc_grad (include/epsm.h: epsm_manifold_grad).
//
// One lane = one light path (wave64, 256-thread workgroups).  Inputs are read
// straight from the reference's own tensor layout ((N,3) fp32 rows: consecutive
// lanes read consecutive 12-byte rows, i.e. fully coalesced dwordx3 streams);
// nothing is staged through HBM between the constraint Jacobian, the block
// solve and the gradient: the per-path state lives in VGPRs (epsm_path_core.h).
// MFMA is deliberately unused: the per-path systems are 2x2 blocks on a band of
// at most 5, not a dense contraction.
#include <string.h>

#include "epsm_common.h"
#include "epsm_path_core.h"

using namespace epsm;

namespace epsm_host {

char *err_buf() {
    static thread_local char buf[512] = "";
    return buf;
}
int fail(int code, const char *what, const char *detail) {
    snprintf(err_buf(), 512, "%s%s%s", what, detail[0] ? ": " : "", detail);
    return code;
}
int hip_fail(const char *what, hipError_t e) {
    const bool nodev = e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorNoBinaryForGpu ||
                       e == hipErrorInsufficientDriver;
    return fail(nodev ? EPSM_ENODEV : EPSM_ELAUNCH, what, hipGetErrorString(e));
}

}  // namespace epsm_host

using epsm_host::fail;

namespace {

constexpr int kBlock = 256;

template <int K, int VARIANT, bool FULL_D>
__global__ __launch_bounds__(kBlock) void epsm_grad_kernel(GradArgs<float> A, int dcols) {
    const int64_t i = (int64_t) blockIdx.x * kBlock + threadIdx.x;
    if (i >= A.N) return;
    const DenseOut<float> out{A, i};
    if (VARIANT == EPSM_VARIANT_MANIFOLD)
        manifold_path<float, K, FULL_D>(A, i, dcols, out);
    else
        caustic_path<float, K, FULL_D>(A, i, dcols, out);
}

template <int K, int VARIANT, bool FULL_D>
hipError_t launch(const GradArgs<float> &A, int dcols, hipStream_t s) {
    const int64_t blocks = (A.N + kBlock - 1) / kBlock;
    hipLaunchKernelGGL((epsm_grad_kernel<K, VARIANT, FULL_D>), dim3((unsigned) blocks), dim3(kBlock), 0, s, A, dcols);
    return hipGetLastError();
}

template <int VARIANT, bool FULL_D>
hipError_t launch_k(int K, const GradArgs<float> &A, int dcols, hipStream_t s) {
    switch (K) {
        case 1: return launch<1, VARIANT, FULL_D>(A, dcols, s);
        case 2: return launch<2, VARIANT, FULL_D>(A, dcols, s);
        case 3: return launch<3, VARIANT, FULL_D>(A, dcols, s);
        case 4: return launch<4, VARIANT, FULL_D>(A, dcols, s);
        default: return launch<5, VARIANT, FULL_D>(A, dcols, s);
    }
}

}  // namespace

extern "C" {

int epsm_abi_version(void) { return EPSM_ABI_VERSION; }

const char *epsm_last_error(void) { return epsm_host::err_buf(); }

int epsm_num_param_grads(int variant, int K) {
    return variant == EPSM_VARIANT_MANIFOLD_CAUSTIC ? 5 * K - 2 : 5 * K;
}

int epsm_manifold_grad(int variant, int64_t N, int K,
                       const float *cam, const EpsmVertexRecord *verts,
                       const float *dlduv, int64_t dlduv_stride, int dlduv_cols,
                       const float *dldp, float clip,
                       float *out_param, float *out_light, float *out_diffuse,
                       void *stream) {
    epsm_host::err_buf()[0] = 0;
    if (variant != EPSM_VARIANT_MANIFOLD && variant != EPSM_VARIANT_MANIFOLD_CAUSTIC)
        return fail(EPSM_EINVAL, "epsm_manifold_grad: unknown variant");
    if (K < 1 || K > EPSM_MAX_VERTICES) return fail(EPSM_EINVAL, "epsm_manifold_grad: K must be in 1..5");
    if (N < 0 || (N + kBlock - 1) / kBlock > 0x7fffffffLL) return fail(EPSM_EINVAL, "epsm_manifold_grad: bad N");
    if (N == 0) return EPSM_OK;   /* empty wavefront: nothing to read or write */
    if (!cam || !verts || !dlduv || !dldp || !out_param || !out_light || !out_diffuse)
        return fail(EPSM_EINVAL, "epsm_manifold_grad: NULL argument");
    if (dlduv_cols < 0 || dlduv_stride < (dlduv_cols < 2 * K ? dlduv_cols : 2 * K))
        return fail(EPSM_EINVAL, "epsm_manifold_grad: dlduv_stride smaller than the columns to read");
    GradArgs<float> A;
    memset(&A, 0, sizeof(A));
    A.N = N;
    A.cam = cam;
    for (int k = 0; k < K; ++k) {
        const EpsmVertexRecord &v = verts[k];
        if (!v.p0 || !v.p1 || !v.p2 || !v.n0 || !v.n1 || !v.n2 || !v.b0 || !v.b1 || !v.eta || !v.light ||
            !v.bsdf || !v.active || !v.active_em || !v.ismesh)
            return fail(EPSM_EINVAL, "epsm_manifold_grad: NULL pointer in a vertex record");
        VertexPtrs<float> &o = A.v[k];
        o.p0 = (const float *) v.p0; o.p1 = (const float *) v.p1; o.p2 = (const float *) v.p2;
        o.n0 = (const float *) v.n0; o.n1 = (const float *) v.n1; o.n2 = (const float *) v.n2;
        o.b0 = (const float *) v.b0; o.b1 = (const float *) v.b1; o.eta = (const float *) v.eta;
        o.light = (const float *) v.light;
        o.bsdf = v.bsdf; o.active = v.active; o.active_em = v.active_em; o.ismesh = v.ismesh;
    }
    A.dlduv = dlduv;
    A.dlduv_stride = dlduv_stride;
    A.dldp = dldp;
    A.clip = (clip > 0.0f && clip <= 3.402823466e+38f) ? clip : 3.402823466e+38f;
    A.out_param = out_param;
    A.out_light = out_light;
    A.out_diffuse = out_diffuse;
    int dcols = dlduv_cols > 2 * K ? 2 * K : dlduv_cols;
    const bool full_d = dcols > 2;
    hipStream_t s = (hipStream_t) stream;
    hipError_t e;
    if (variant == EPSM_VARIANT_MANIFOLD)
        e = full_d ? launch_k<EPSM_VARIANT_MANIFOLD, true>(K, A, dcols, s)
                   : launch_k<EPSM_VARIANT_MANIFOLD, false>(K, A, dcols, s);
    else
        e = full_d ? launch_k<EPSM_VARIANT_MANIFOLD_CAUSTIC, true>(K, A, dcols, s)
                   : launch_k<EPSM_VARIANT_MANIFOLD_CAUSTIC, false>(K, A, dcols, s);
    if (e != hipSuccess) return epsm_host::hip_fail("epsm_manifold_grad", e);
    return EPSM_OK;
}

}  // extern "C"
