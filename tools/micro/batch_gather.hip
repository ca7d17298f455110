// Round 5, VERDICT r4 item 3: can the backward kernel's record reads be made WHOLE-LINE requests within its LDS budget?
// Every lane of a wave needs ONE 128-byte record (records (N, K = 5, 32 words); the wave's 64 records scattered over a 256-path
// window; two records per path: vertices 0 and 1), consumes it (WORK dependent FMAs per quad), no prefetch -- the shape of the
// kernel's round -- at THREE waves per SIMD (256 threads, <= 53 KB of LDS per workgroup):
//   mode 0  per-lane loads: six 16-byte loads per lane now, quads 6 and 7 after the work (the kernel's geo / addr stages)
//   mode 1  eight LDS-DMA instructions per wave, eight lanes per record, 8 KB of staging per wave (what does NOT fit beside the
//           kernel's table: the upper bound of the idea)
//   mode 2  the same in BATCHES of 16 records through 2 304 bytes of staging per wave (what would fit in the wave's row queue):
//           two DMA instructions, wait, the 16 owning lanes read their eight quads, next batch
//   mode 3  batches of 16 records, double-buffered (4 608 bytes per wave): batch b + 1 in flight while batch b is read back
// hipcc -O3 --offload-arch=gfx950 -o tools/micro/batch_gather tools/micro/batch_gather.hip && tools/micro/batch_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float F4 __attribute__((ext_vector_type(4)));
constexpr int K = 5;
constexpr int kSlot = 288;                 // floats per DMA instruction's slot: 1024 bytes + 128 of padding (conflict-free read-back)

template <int WORK> __device__ __forceinline__ float consume(const F4 *q, int n, float acc) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (j >= n) break;
        float v = q[j].x + q[j].y + q[j].z + q[j].w;
#pragma unroll 8
        for (int w = 0; w < WORK; ++w) v = fmaf(v, 1.0001f, acc);
        acc += v;
    }
    return acc;
}
__device__ __forceinline__ void dma16(const float *src, float *dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) src, (__attribute__((address_space(3))) void *) dst, 16, 0, 0);
}

template <int MODE, int WORK>
__global__ __launch_bounds__(256, 3) void k(const float *rec, float *out, int64_t n, int steps, int windows_per_block) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float *stage = lds + wv * (MODE == 1 ? 8 * kSlot : MODE == 2 ? 2 * kSlot : 4 * kSlot);
    float acc = 0.f;
    for (int wi = 0; wi < windows_per_block; ++wi) {
        const int64_t base = ((int64_t) blockIdx.x * windows_per_block + wi) * 256;
        if (base >= n) break;
        const unsigned r = __brev(threadIdx.x) >> 24;      // this lane's path inside the window
        const int64_t path = base + r;
        for (int s = 0; s < steps; ++s) {
            F4 cur[8];
            const float *mine = rec + (path * K + s) * 32;
            if (MODE == 0) {
#pragma unroll
                for (int j = 0; j < 6; ++j) cur[j] = *(const F4 *) (mine + 4 * j);
                acc = consume<WORK>(cur, 6, acc);
                cur[6] = *(const F4 *) (mine + 24); cur[7] = *(const F4 *) (mine + 28);
                acc = consume<0>(cur + 6, 2, acc);
                continue;
            }
            // instruction j of a group, lane l: record of lane 8 j + l / 8 (of the batch), source quad (l % 8) ^ (owner % 8)
            auto issue = [&](int first_owner, int ninstr, float *dst) {
                for (int j = 0; j < ninstr; ++j) {
                    const int owner = first_owner + 8 * j + (lane >> 3);
                    const unsigned ro = __brev((unsigned) (wv * 64 + owner)) >> 24;
                    const int quad = (lane & 7) ^ (owner & 7);
                    dma16(rec + ((base + ro) * K + s) * 32 + 4 * quad, dst + j * kSlot);
                }
            };
            // the owning lanes read their record back: slot (lane_in_batch / 8), record lane % 8, quad q at position q ^ (lane % 8)
            auto readback = [&](int first_owner, int ninstr, const float *src) {
                const int li = lane - first_owner;
                if (li >= 0 && li < 8 * ninstr) {
                    const float *p = src + (li >> 3) * kSlot + (li & 7) * 32;
#pragma unroll
                    for (int q = 0; q < 8; ++q) cur[q] = *(const F4 *) (p + 4 * (q ^ (li & 7)));
                }
            };
            if (MODE == 1) {
                issue(0, 8, stage);
                __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
                readback(0, 8, stage);
            } else if (MODE == 2) {
#pragma unroll 1
                for (int b = 0; b < 4; ++b) {
                    issue(16 * b, 2, stage);
                    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    readback(16 * b, 2, stage);
                    __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
            } else {
                issue(0, 2, stage);
#pragma unroll 1
                for (int b = 0; b < 4; ++b) {
                    if (b + 1 < 4) {
                        issue(16 * (b + 1), 2, stage + ((b + 1) & 1) * 2 * kSlot);
                        __asm__ volatile("s_waitcnt vmcnt(2)" ::: "memory");
                    } else {
                        __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                    readback(16 * b, 2, stage + (b & 1) * 2 * kSlot);
                    __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
            }
            __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            acc = consume<WORK>(cur, 6, acc);
            acc = consume<0>(cur + 6, 2, acc);
        }
    }
    out[(int64_t) blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE, int WORK> float run(const float *rec, float *out, int64_t n, int steps) {
    const int blocks = 8192, per = (int) ((n / 256 + blocks - 1) / blocks);
    const int lds = 4 * 4 * (MODE == 1 ? 8 * kSlot : MODE == 2 ? 2 * kSlot : MODE == 3 ? 4 * kSlot : 64);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipFuncSetAttribute((const void *) k<MODE, WORK>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    k<MODE, WORK><<<blocks, 256, lds>>>(rec, out, n, steps, per);
    hipEventRecord(a);
    for (int rep = 0; rep < 3; ++rep) k<MODE, WORK><<<blocks, 256, lds>>>(rec, out, n, steps, per);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / 3;
}

template <int WORK> void row(const float *rec, float *out, int64_t n, int steps) {
    const float t0 = run<0, WORK>(rec, out, n, steps), t1 = run<1, WORK>(rec, out, n, steps), t2 = run<2, WORK>(rec, out, n, steps),
                t3 = run<3, WORK>(rec, out, n, steps);
    const double gb = n * steps * 128 / 1e9;
    printf("steps %d, %3d FMAs/quad: per-lane %.3f ms (%.2f TB/s)   DMA 8 KB/wave %.3f (%.2f)   DMA batches of 16, 2.3 KB/wave %.3f (%.2f)   double-buffered 4.6 KB/wave %.3f (%.2f)\n",
           steps, WORK, t0, gb / t0, t1, gb / t1, t2, gb / t2, t3, gb / t3);
}

int main() {
    const int64_t n = 1ll << 24;
    float *rec, *out;
    if (hipMalloc(&rec, n * K * 128) != hipSuccess || hipMalloc(&out, 4 * 8192 * 256) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(rec, 0, n * K * 128);
    for (int steps : {1, 2}) {
        row<0>(rec, out, n, steps);
        row<64>(rec, out, n, steps);
        row<192>(rec, out, n, steps);
    }
    return 0;
}
