// Micro-benchmark: rate of LDS float atomics / CAS / integer atomics / plain read-modify-write with scattered addresses
// (the access pattern of LdsTable::add in epsm_wave_scatter.h).
//   hipcc --offload-arch=gfx950 -O3 -o lds_atomics lds_atomics.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
constexpr int kRows = 2048, kIters = 4096;
template <int MODE> __global__ __launch_bounds__(256) void k(float *out, uint32_t seed) {
    __shared__ float vals[kRows * 3];
    __shared__ uint32_t keys[kRows];
    for (int e = threadIdx.x; e < kRows; e += 256) { keys[e] = 0xFFFFFFFFu; vals[3 * e] = vals[3 * e + 1] = vals[3 * e + 2] = 0.f; }
    __syncthreads();
    uint32_t x = seed + blockIdx.x * 256 + threadIdx.x;
    float acc = 0.f;
    for (int it = 0; it < kIters; ++it) {
        x = x * 1664525u + 1013904223u;
        const uint32_t slot = ((MODE == 4 || MODE == 8 || MODE == 9 || MODE == 10 || MODE == 12) ? (x >> 8) & 15u : (x >> 8)) & (kRows - 1);     // MODE 4, 8, 9, 10: 16 hot rows
        if (MODE == 0) { atomicAdd(&vals[3 * slot], 1.f); atomicAdd(&vals[3 * slot + 1], 2.f); atomicAdd(&vals[3 * slot + 2], 3.f); }
        if (MODE == 1) { const uint32_t p = atomicCAS(&keys[slot], 0xFFFFFFFFu, slot); acc += p; }
        if (MODE == 2) { vals[3 * slot] += 1.f; vals[3 * slot + 1] += 2.f; vals[3 * slot + 2] += 3.f; }     // racy plain RMW: rate only
        if (MODE == 3 || MODE == 4) { const uint32_t p = atomicCAS(&keys[slot], 0xFFFFFFFFu, slot); acc += p;
                         atomicAdd(&vals[3 * slot], 1.f); atomicAdd(&vals[3 * slot + 1], 2.f); atomicAdd(&vals[3 * slot + 2], 3.f); }
        if (MODE == 5 || MODE == 8) {                                  // float add as a compare-and-swap loop
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                uint32_t *a = (uint32_t *) &vals[3 * slot + c];
                uint32_t old = *(volatile uint32_t *) a, assumed;
                do { assumed = old; old = atomicCAS(a, assumed, __float_as_uint(__uint_as_float(assumed) + 1.f + c)); } while (old != assumed);
            }
        }
        if (MODE == 6 || MODE == 9) { uint32_t *a = (uint32_t *) vals; atomicAdd(&a[3 * slot], 1u); atomicAdd(&a[3 * slot + 1], 2u); atomicAdd(&a[3 * slot + 2], 3u); }
        if (MODE == 7 || MODE == 10) { unsigned long long *a = (unsigned long long *) vals; const uint32_t s2 = slot & 511u;
                         atomicAdd(&a[3 * s2], 1ull); atomicAdd(&a[3 * s2 + 1], 2ull); atomicAdd(&a[3 * s2 + 2], 3ull); }
        if (MODE == 11 || MODE == 12) { double *a = (double *) vals; const uint32_t s2 = slot & 511u;      // ds_add_f64 (gfx90a+)
                         atomicAdd(&a[3 * s2], 1.0); atomicAdd(&a[3 * s2 + 1], 2.0); atomicAdd(&a[3 * s2 + 2], 3.0); }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = vals[0] + acc;
}
template <int MODE> void run(const char *name, int lane_ops) {
    float *out; hipMalloc(&out, 4096 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256 * 8;
    k<MODE><<<blocks, 256>>>(out, 1); hipDeviceSynchronize();
    hipEventRecord(a); k<MODE><<<blocks, 256>>>(out, 2); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double ops = (double) blocks * 256 * kIters * lane_ops;
    printf("%-40s %8.3f ms  %8.2f G lane-ops/s = %5.2f lane-ops / clock / CU (256 CUs, 2.4 GHz)\n", name, ms, ops / ms / 1e6, ops / (ms * 1e-3) / 256 / 2.4e9);
    hipFree(out);
}
int main() {
    run<0>("3 x ds_add_f32, scattered", 3);
    run<1>("ds_cmpst_rtn, scattered", 1);
    run<2>("3 x plain read+add+write, scattered", 3);
    run<3>("cmpst + 3 x ds_add_f32, scattered", 4);
    run<4>("cmpst + 3 x ds_add_f32, 16 hot rows", 4);
    run<5>("3 x float add by CAS loop, scattered", 3);
    run<8>("3 x float add by CAS loop, 16 hot rows", 3);
    run<6>("3 x ds_add_u32, scattered", 3);
    run<7>("3 x ds_add_u64, scattered", 3);
    run<9>("3 x ds_add_u32, 16 hot rows", 3);
    run<10>("3 x ds_add_u64, 16 hot rows", 3);
    run<11>("3 x ds_add_f64, scattered", 3);
    run<12>("3 x ds_add_f64, 16 hot rows", 3);
    return 0;
}
