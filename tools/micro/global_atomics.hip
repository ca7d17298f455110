// Rate of the flush's global float atomics on MI355X: groups of 4 lanes add the x,y,z of one 12-byte row (one request per
// row, as LdsTable::flush issues them) to rows drawn at random from a buffer of R rows.
// hipcc -O3 --offload-arch=gfx950 -o /tmp/global_atomics tools/micro/global_atomics.hip && /tmp/global_atomics
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ __launch_bounds__(256) void k(float *buf, uint32_t rows, int iters, int spread) {
    uint32_t s = (blockIdx.x * 256u + threadIdx.x) >> 2;
    const int c = threadIdx.x & 3;
    s = s * 747796405u + 2891336453u;
    for (int i = 0; i < iters; ++i) {
        s = s * 1664525u + 1013904223u;
        uint32_t r = (uint32_t) (((unsigned long long) (s ^ (s >> 15)) * rows) >> 32);
        if (spread == 0) r = (blockIdx.x * 64u + (threadIdx.x >> 2) + 64u * 2048u * i) % rows;   // no two workgroups on one row
        if (c < 3) atomicAdd(buf + 3ull * r + c, 1.0f);
    }
}
int main() {
    const int blocks = 2048, iters = 64;
    float *buf; hipMalloc(&buf, 3ull * 4 * 16000000); hipMemset(buf, 0, 3ull * 4 * 16000000);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const uint32_t R[] = {1, 16, 96, 1024, 15658, 200000, 2000000, 16000000};
    for (int spread = 1; spread >= 0; --spread)
        for (uint32_t rows : R) {
            if (!spread && rows < 8388608u) continue;
            k<<<blocks, 256>>>(buf, rows, iters, spread);
            hipEventRecord(a);
            for (int rep = 0; rep < 5; ++rep) k<<<blocks, 256>>>(buf, rows, iters, spread);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
            const double n = (double) blocks * 64 * iters;
            printf("%s rows %9u: %8.3f ms for %.2e row-atomics = %7.2f G rows/s (%.1f ns per atomic on one row if serial)\n",
                   spread ? "random " : "disjoint", rows, ms, n, n / ms * 1e-6, ms * 1e6 / (n / rows));
        }
    return 0;
}
