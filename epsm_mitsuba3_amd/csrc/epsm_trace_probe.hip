// epsm_trace_probe.hip -- epsm_probe (include/epsm_trace.h): the tracer's per-path functions on plain numbers, on the device.
#include "epsm_common.h"
#include "../../include/epsm_trace.h"
#include "epsm_probe_core.h"

using namespace epsm;
using epsm_host::fail;

namespace {
__global__ __launch_bounds__(256) void epsm_probe_kernel(int what, int64_t n, const float *in, float *out, EpsmBsdf bsdf, EpsmSensor sensor) {
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i < n) probe_row(what, in + i * EPSM_PROBE_IN, out + i * EPSM_PROBE_OUT, &bsdf, &sensor);
}
}  // namespace

extern "C" int epsm_probe(int what, int64_t n, const float *in, float *out, const void *cfg, void *stream) {
    epsm_host::err_buf()[0] = 0;
    if (what < 0 || what >= EPSM_PROBE_COUNT) return fail(EPSM_EINVAL, "epsm_probe: unknown function");
    if (n == 0) return EPSM_OK;
    if (n < 0 || !in || !out) return fail(EPSM_EINVAL, "epsm_probe: bad argument");
    if ((probe_needs_bsdf(what) || probe_needs_sensor(what)) && !cfg) return fail(EPSM_EINVAL, "epsm_probe: this function needs cfg");
    EpsmBsdf bsdf = {};
    EpsmSensor sensor = {};
    if (probe_needs_bsdf(what)) bsdf = *(const EpsmBsdf *) cfg;
    if (probe_needs_sensor(what)) sensor = *(const EpsmSensor *) cfg;
    hipLaunchKernelGGL(epsm_probe_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, (hipStream_t) stream, what, n, in, out,
                       bsdf, sensor);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return epsm_host::hip_fail("epsm_probe", e);
    return EPSM_OK;
}
