// Rate of v_fma_f32 against v_pk_fma_f32 on MI355X: 16 independent chains per lane, ITER turns, one wave..eight waves per SIMD.
// hipcc -O3 --offload-arch=gfx950 -o tools/micro/pk_fma tools/micro/pk_fma.hip && tools/micro/pk_fma
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float float2v __attribute__((ext_vector_type(2)));
template <bool PACKED>
__global__ __launch_bounds__(256) void k(float *out, float a, float b, int iters) {
    float2v acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = float2v{(float) threadIdx.x + j, (float) j};
    const float2v av = {a, a * 1.0001f}, bv = {b, b * 0.9999f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (PACKED) {
                asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(acc[j]) : "v"(av), "v"(bv));
            } else {
                asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(acc[j].x) : "v"(av.x), "v"(bv.x));
                asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(acc[j].y) : "v"(av.y), "v"(bv.y));
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += acc[j].x + acc[j].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float *out; hipMalloc(&out, 4 * 256 * 8192);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096;
    for (int blocks : {256, 1024, 2048, 4096}) {
        for (int packed = 0; packed < 2; ++packed) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (packed) k<true><<<blocks, 256>>>(out, 0.999f, 0.001f, iters); else k<false><<<blocks, 256>>>(out, 0.999f, 0.001f, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double fma = (double) blocks * 256 * iters * 16;
            printf("%d workgroups of 256 (%.1f waves per SIMD) %s: %.3f ms, %.1f TFLOP/s\n", blocks, blocks * 4 / 1024.0,
                   packed ? "v_pk_fma_f32" : "v_fma_f32   ", ms, 2 * fma / ms * 1e-9);
        }
    }
    return 0;
}
