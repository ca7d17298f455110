// Two ways for the lanes of a wave to get one 128-byte record each (records (N, K=5, 32 words), a wave's 64 paths
// scattered over a 256-path window as in epsm_backward_pass_packed), at two waves per SIMD, STEPS vertices per path,
// next vertex prefetched while the current one is consumed (WORK dependent FMAs per quad word):
//   A: eight 16-byte loads per lane (64 lanes x 64 different records per instruction),
//   B: eight LDS-DMA instructions per wave, EIGHT LANES PER RECORD (a wave instruction covers eight whole records),
//      then eight ds_read_b128 per lane from the wave's 8 KB staging area.
// hipcc -O3 --offload-arch=gfx950 -o tools/micro/coop_gather tools/micro/coop_gather.hip && tools/micro/coop_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float F4 __attribute__((ext_vector_type(4)));
constexpr int K = 5;

template <int WORK> __device__ __forceinline__ float consume(const F4 q[8], float acc) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float v = q[j].x + q[j].y + q[j].z + q[j].w;
#pragma unroll
        for (int w = 0; w < WORK; ++w) v = fmaf(v, 1.0001f, acc);
        acc += v;
    }
    return acc;
}

template <int MODE, int WORK>
__global__ __launch_bounds__(256, 2) void k(const float *rec, float *out, int64_t n, int steps, int windows_per_block) {
    extern __shared__ float lds[];                         // 80 KB: two workgroups per CU; the first 32 KB are staging
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float *stage = lds + wv * 2048;                        // 8 KB per wave
    float acc = 0.f;
    for (int wi = 0; wi < windows_per_block; ++wi) {
        const int64_t base = ((int64_t) blockIdx.x * windows_per_block + wi) * 256;
        if (base >= n) break;
        const unsigned r = __brev(threadIdx.x) >> 24;      // this lane's path inside the window
        const int64_t path = base + r;
        F4 cur[8], nxt[8];
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) nxt[j] = *(const F4 *) (rec + (path * K + 0) * 32 + 4 * j);
            for (int s = 0; s < steps; ++s) {
#pragma unroll
                for (int j = 0; j < 8; ++j) cur[j] = nxt[j];
                if (s + 1 < steps) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) nxt[j] = *(const F4 *) (rec + (path * K + s + 1) * 32 + 4 * j);
                }
                acc = consume<WORK>(cur, acc);
            }
        } else {
            // instruction j, lane l: record of the wave's lane 8 j + l / 8, quad (l % 8 + that lane) % 8 (rotated: conflict-free read-back)
            auto issue = [&](int s) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int owner = 8 * j + (lane >> 3);
                    const unsigned ro = __brev((unsigned) (wv * 64 + owner)) >> 24;
                    const int64_t p = base + ro;
                    const int quad = ((lane & 7) - owner) & 7;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (rec + (p * K + s) * 32 + 4 * quad),
                                                     (__attribute__((address_space(3))) void *) (stage + j * 256), 16, 0, 0);
                }
            };
            issue(0);
            for (int s = 0; s < steps; ++s) {
                __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int j = 0; j < 8; ++j)                   // own record: 128 B at lane * 128, quad j stored at slot (j + lane) % 8
                    cur[j] = *(const F4 *) (stage + lane * 32 + 4 * ((j + lane) & 7));
                __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (s + 1 < steps) issue(s + 1);
                acc = consume<WORK>(cur, acc);
            }
        }
    }
    out[(int64_t) blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE, int WORK> float run(const float *rec, float *out, int64_t n, int steps) {
    const int blocks = 2048, per = (int) ((n / 256 + blocks - 1) / blocks);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipFuncSetAttribute((const void *) k<MODE, WORK>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    k<MODE, WORK><<<blocks, 256, 80 * 1024>>>(rec, out, n, steps, per);
    hipEventRecord(a);
    for (int rep = 0; rep < 3; ++rep) k<MODE, WORK><<<blocks, 256, 80 * 1024>>>(rec, out, n, steps, per);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / 3;
}

int main() {
    const int64_t n = 1ll << 24;
    float *rec, *out;
    if (hipMalloc(&rec, n * K * 128) != hipSuccess || hipMalloc(&out, 4 * 2048 * 256) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(rec, 0, n * K * 128);
    for (int steps : {1, 3, 5}) {
        printf("steps %d, no work:   per-lane loads %.3f ms   eight lanes per record through LDS %.3f ms   (%.2f GB)\n", steps,
               run<0, 0>(rec, out, n, steps), run<1, 0>(rec, out, n, steps), n * steps * 128 / 1e9);
        printf("steps %d, 16 FMAs/word: per-lane loads %.3f ms   eight lanes per record through LDS %.3f ms\n", steps,
               run<0, 16>(rec, out, n, steps), run<1, 16>(rec, out, n, steps));
    }
    return 0;
}
