// Calibration of rocprofv3's FETCH_SIZE for the access pattern of epsm_backward_pass_packed: every lane reads ONE
// 128-byte record with eight 16-byte loads, the 64 lanes of a wave taking records scattered over a 256-record window
// (here: bit-reversed order inside the window).  Known traffic: N x 128 B read, N x 4 B written.
//   hipcc -O3 --offload-arch=gfx950 -o gather128 gather128.hip && rocprofv3 --pmc FETCH_SIZE -- ./gather128
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef float F4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void gather128_kernel(const float *rec, float *out, int64_t n, int stride_recs, int touch) {
    const int64_t t = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const unsigned r = __brev(threadIdx.x) >> 24;                       // permutation inside the 256-record window
    const int64_t i = ((t & ~255ll) | r) * stride_recs;
    const F4 *p = (const F4 *) (rec + i * 32);
    if (touch) {                                                        // round 4's pattern: ONE word of the line first (it pulls the
        const float w = *(const volatile float *) (rec + i * 32);       // line into L2), the eight quads afterwards (L2 hits)
        if (w == 123.f) out[t] = w;
        __builtin_amdgcn_s_sleep(64);
    }
    F4 s = p[0];
#pragma unroll
    for (int q = 1; q < 8; ++q) s += p[q];
    out[t] = s.x + s.y + s.z + s.w;
}

int main(int argc, char **argv) {
    const int64_t n = 1ll << 23;                                        // 8 M records = 1 GB (x stride)
    const int stride = argc > 1 ? atoi(argv[1]) : 1;                    // 5: only every 5th record is read (K = 5, one live vertex)
    const int touch = argc > 2 ? atoi(argv[2]) : 0;                     // 1: a one-word load of every record before its eight quads
    float *rec, *out;
    if (hipMalloc(&rec, n * stride * 128) != hipSuccess || hipMalloc(&out, n * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(rec, 0, n * stride * 128);
    for (int rep = 0; rep < 5; ++rep)
        hipLaunchKernelGGL(gather128_kernel, dim3((unsigned) (n / 256)), dim3(256), 0, 0, rec, out, n, stride, touch);
    hipDeviceSynchronize();
    printf("gather128: %lld records of 128 B (stride %d, touch %d) = %.3f GB read per launch\n", (long long) n, stride, touch, n * 128 / 1e9);
    return 0;
}
